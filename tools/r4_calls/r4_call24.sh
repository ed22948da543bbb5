# Round 4, GPU call 24: PMC passes on a run in rounds (C3 with drift, 2 M users, one simulation per process): what the sweep's and
# k_advance_run's waves do with their cycles (SQ), LDS bank conflicts and matrix-pipe busy cycles, bytes at the L2 <-> fabric
# boundary.  Each counter group in its own run, kernel-trace only beside it.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pmc() { # name, counters, bench args
  name=$1; cnt=$2; shift; shift
  timeout 400 rocprofv3 --kernel-trace --pmc $cnt --output-format csv -d $O/$name -o run -- python $R/bench.py "$@" --single-run > $O/$name.out 2> $O/$name.err
  f=$(find $O/$name -name '*counter_collection.csv' | head -1)
  python - "$f" "$O/${name}_counters.csv" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if (k, r['Dispatch_Id']) not in seen:
        seen.add((k, r['Dispatch_Id'])); n[k] += 1; dur[k] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
with open(sys.argv[2], 'w') as f:
    f.write('kernel,dispatches,total_ns,counter,value\n')
    for k in sorted(agg, key=lambda k: -dur[k]):
        for c, v in sorted(agg[k].items()):
            f.write(f'"{k[:90]}",{n[k]},{dur[k]},{c},{v:.6g}\n')
PY
  grep '"metric"' $O/$name.out > $O/${name}_bench_line.json
  rm -rf $O/$name
  head -12 $O/${name}_counters.csv | cut -c1-200
}
A="--workload c3drift --users 2000000 --no-cpu-baseline --no-drift-line --no-materialise"
pmc pmc_c3drift_rounds_sq "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" $A
pmc pmc_c3drift_rounds_mfma "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" $A
pmc pmc_c3drift_rounds_fetch FETCH_SIZE $A
pmc pmc_c3drift_rounds_write WRITE_SIZE $A
