"""Instructions a kernel spends per source line (hipcc -gline-tables-only -save-temps on one unit):
    python tools/line_cost.py <part 1..8> <regex of the mangled kernel name> [first_line last_line]
prints the count per source line in the range (or the 40 heaviest lines), split into vector / scalar / LDS / memory."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNITS = ['rg_host', 'rg_exact', 'rg_draw_fp32', 'rg_draw_pipelined', 'rg_draw_wide', 'rg_advance', 'rg_walk', 'rg_draw_exacthi']
part, pat = sys.argv[1], sys.argv[2]
unit = UNITS[int(part) - 1]
SRC = os.path.join(ROOT, 'recogym_amd', 'csrc', unit + '.hip')
tmp = tempfile.mkdtemp()
subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-gline-tables-only',
                '-save-temps', '-c', '-o', 'x.o', SRC], cwd=tmp, stderr=subprocess.DEVNULL, check=True)
txt = open(os.path.join(tmp, unit + '-hip-amdgcn-amd-amdhsa-gfx950.s')).read()
m = re.search(r'^(_Z\w*' + pat + r'\w*):(.*?)\.Lfunc_end', txt, re.S | re.M)
print('kernel', m.group(1))
files = {int(a): b for a, b in re.findall(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', txt)}
files.update({int(a): b for a, b in re.findall(r'\.file\s+(\d+)\s+"([^"]+)"\s*$', txt, re.M) if int(a) not in files})
src = open(SRC).read().split('\n')
cur = None
cnt = collections.defaultdict(lambda: collections.Counter())
tot = collections.Counter()
for l in m.group(2).split('\n'):
    mm = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if mm:
        cur = (int(mm.group(1)), int(mm.group(2)))
        continue
    t = l.strip()
    if not t or t.startswith(('.', ';', '//')) or t.endswith(':'):
        continue
    op = t.split()[0]
    kind = 'v' if op.startswith('v_') else 's' if op.startswith('s_') else 'lds' if op.startswith('ds_') else 'mem'
    cnt[cur][kind] += 1
    tot[kind] += 1
print('total', dict(tot))
main = [k for k, v in files.items() if v.endswith(unit + '.hip')]
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (None, None)
rows = []
for (f, l), c in cnt.items():
    name = files.get(f, '?')
    rows.append((name.endswith(unit + '.hip'), l, sum(c.values()), dict(c), os.path.basename(name)))
if lo is not None:
    sel = [r for r in rows if r[0] and lo <= r[1] <= hi]
    s = collections.Counter()
    for r in sorted(sel, key=lambda r: r[1]):
        print(f'{r[2]:5d} {str(r[3]):48s} line {r[1]:5d}  {src[r[1] - 1].strip()[:90]}')
        s.update(r[3])
    print('range total', dict(s), sum(s.values()))
else:
    for r in sorted(rows, key=lambda r: -r[2])[:40]:
        print(f'{r[2]:5d} {str(r[3]):48s} {r[4]}:{r[1]:5d}  {src[r[1] - 1].strip()[:80] if r[0] else ""}')
