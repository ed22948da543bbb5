# Per-dispatch SQ counters of the walk rounds on a 2 M-user C3 run (rocprofv3 --pmc in its own pass, --kernel-trace only)
mkdir -p gpurun_out/r2_pmc
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
RECOGYM_DEBUG=1 timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/r2_pmc/sq -- python $R/bench.py --workload c3 --users 2000000 --steps 1 --warmup 0 --no-cpu-baseline --no-drift-line > $R/gpurun_out/r2_pmc/sq.out 2> $R/gpurun_out/r2_pmc/sq.err
grep "walk round" $R/gpurun_out/r2_pmc/sq.err | tail -2
cd $R/gpurun_out/r2_pmc
python - <<'PY'
import csv, glob, collections
fs = glob.glob('sq/**/*counter_collection.csv', recursive=True)
rows = collections.defaultdict(dict)
for r in csv.DictReader(open(fs[0])):
    if 'k_walk<' in r['Kernel_Name']:
        rows[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
        rows[int(r['Dispatch_Id'])]['ns'] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for k in sorted(rows):
    v = rows[k]
    print(k, {a: f'{b:.4g}' for a, b in v.items()})
PY
rm -rf sq
