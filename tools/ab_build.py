"""A/B builds of ONE translation unit with compile-time switches, linked against the default build's other objects.
    python tools/ab_build.py <unit index 1..9> name1:-DFLAG=1 name2:-DA=2,-DB=3 ...
-> recogym_amd/csrc/librecogym_hip_<name>.so (load with RECOGYM_HIP_LIB)."""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402
g.build()
csrc = os.path.join(ROOT, 'recogym_amd', 'csrc')
unit = int(sys.argv[1])
procs = []
for spec in sys.argv[2:]:
    name, flags = spec.split(':', 1)
    objdir = os.path.join(csrc, 'build_librecogym_hip_' + name)
    os.makedirs(objdir, exist_ok=True)
    for p in range(1, g.N_PARTS + 1):
        if p != unit:
            shutil.copy2(os.path.join(csrc, 'build', f'part{p}.o'), os.path.join(objdir, f'part{p}.o'))
    cmd = [g.HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', *g.UNIT_FLAGS.get(g.UNITS[unit - 1], []), *flags.split(','), '-c', '-o',
           os.path.join(objdir, f'part{unit}.o'), os.path.join(csrc, g.UNITS[unit - 1] + '.hip')]
    procs.append((name, objdir, subprocess.Popen(cmd)))
    if len(procs) % 6 == 0:
        for _, _, pr in procs[-6:]:
            pr.wait()
for name, objdir, pr in procs:
    assert pr.wait() == 0, name
    out = os.path.join(csrc, f'librecogym_hip_{name}.so')
    subprocess.check_call([g.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] +
                          [os.path.join(objdir, f'part{p}.o') for p in range(1, g.N_PARTS + 1)])
    print('built', out)
