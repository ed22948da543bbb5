mkdir -p gpurun_out/r2_trace
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_trace/c3 -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2_trace/c3.out 2> $GRAFT_REPO_ROOT/gpurun_out/r2_trace/c3.err
cd $GRAFT_REPO_ROOT/gpurun_out/r2_trace
python - <<'PY'
import csv, glob, collections
f = glob.glob('c3/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(len(rows), rows[0].keys())
agg = collections.defaultdict(list)
for r in rows:
    agg[r['Kernel_Name'][:60]].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
out = []
for k, v in agg.items():
    d = [e - s for s, e in v]
    out.append((sum(d), k, len(d), min(d), sorted(d)[len(d)//2], max(d)))
for tot, k, n, mn, med, mx in sorted(out, reverse=True)[:16]:
    print(f'{tot/1e6:9.2f} ms  n={n:5d}  min={mn/1e3:8.1f}us med={med/1e3:8.1f}us max={mx/1e3:9.1f}us  {k}')
# compact per-launch dump for offline analysis: kernel id, start, dur
names = sorted(agg)
with open('c3_launches.tsv', 'w') as g:
    for r in rows:
        g.write(f"{names.index(r['Kernel_Name'][:60])}\t{r['Start_Timestamp']}\t{int(r['End_Timestamp'])-int(r['Start_Timestamp'])}\n")
    g.write('# ' + ' | '.join(names) + '\n')
PY
rm -rf c3
