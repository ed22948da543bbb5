"""Where a kernel's register spills are: hipcc -gline-tables-only -save-temps on one translation unit (1 .. 7: rg_host, rg_exact, rg_draw_fp32, rg_draw_pipelined, rg_draw_wide, rg_advance, rg_walk), scratch_* instructions
of the chosen kernel mapped to the source lines they were generated for.
    python tools/spill_map.py <part 1..7> <regex of the mangled kernel name> [loop-start-text loop-end-text]
e.g. python tools/spill_map.py 5 'k_draw_f16wILi32ELi13ELi1E' 'for (uint32_t ti = pt_lo; ti < pt_hi; ++ti) {' "the last pair's own sums"
"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNITS = ['rg_host', 'rg_exact', 'rg_draw_fp32', 'rg_draw_pipelined', 'rg_draw_wide', 'rg_advance', 'rg_walk', 'rg_draw_exacthi', 'rg_draw_lds']
part, pat = sys.argv[1], sys.argv[2]
unit = UNITS[int(part) - 1]
SRC = os.path.join(ROOT, 'recogym_amd', 'csrc', unit + '.hip')
tmp = tempfile.mkdtemp()
subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-gline-tables-only',
                '-save-temps', '-c', '-o', 'x.o', SRC], cwd=tmp, stderr=subprocess.DEVNULL, check=True)
txt = open(os.path.join(tmp, unit + '-hip-amdgcn-amd-amdhsa-gfx950.s')).read()
m = re.search(r'^(_Z\w*' + pat + r'\w*):(.*?)\.Lfunc_end', txt, re.S | re.M)
print('kernel', m.group(1))
src = open(SRC).read().split('\n')
cur, cnt, n = None, collections.Counter(), 0
for l in m.group(2).split('\n'):
    mm = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if mm:
        cur = (int(mm.group(1)), int(mm.group(2)))
        continue
    t = l.strip()
    if not t or t.startswith(('.', ';', '//')) or t.endswith(':'):
        continue
    n += 1
    if 'scratch_' in t:
        cnt[cur] += 1
print(f'{n} instructions, {sum(cnt.values())} scratch loads/stores')
if len(sys.argv) > 4:
    lo = [i for i, l in enumerate(src) if sys.argv[3] in l][-1] + 1
    hi = [i for i, l in enumerate(src) if sys.argv[4] in l][-1] + 1
    print(f'source lines {lo}..{hi}: {sum(v for (f, l), v in cnt.items() if f == 1 and lo <= l < hi)} of them')
for (f, l), v in sorted(cnt.items(), key=lambda x: -x[1])[:12]:
    print(f'{v:5d}  line {l:5d}  {src[l - 1].strip()[:100] if f == 1 and l else "(inlined header / compiler generated)"}')
