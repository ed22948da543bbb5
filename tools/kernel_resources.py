"""Register / scratch / occupancy table of every kernel from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
    hipcc ... -Rpass-analysis=kernel-resource-usage ... 2> remarks.txt ; python tools/kernel_resources.py remarks.txt [filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
rows = []
for b in blocks:
    name = b.split('\n')[0].split(' ')[0]
    def g(k):
        m = re.search(k + r': (\d+)', b)
        return int(m.group(1)) if m else -1
    rows.append((name, g('VGPRs'), g('AGPRs'), g('SGPRs'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'),
                 g('VGPRs Spill'), g(r'LDS Size \[bytes/block\]')))
dem = subprocess.run(['c++filt'] + [r[0] for r in rows], capture_output=True, text=True).stdout.split('\n')
for r, d in zip(rows, dem):
    d = re.sub(r'\(anonymous namespace\)::', '', d).split('(')[0].replace('void ', '')
    if flt in d:
        print(f'{d[:64]:64s} vgpr{r[1]:4d} agpr{r[2]:4d} sgpr{r[3]:4d} scratch{r[4]:5d} occ{r[5]:2d} spill{r[6]:4d} lds{r[7]:6d}')
