# Round-6 profile evidence on the BENCHED configuration (C3, 10 M users — as in round 4):
# kernel-trace stats of the bench command, then FETCH_SIZE / WRITE_SIZE / SQ passes, each in its own run on `bench.py
# --single-run` (one simulation per process).  Outputs under gpurun_out/r6_prof/ (tools/make_pmc_traffic.py r6 copies the
# summaries to profiles/r6/ and derives pmc_traffic.json).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stats() { # name, bench args
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o run -- python $R/bench.py "$@" > $O/$name.bench.log 2> $O/$name.err
  f=$(find $O/$name -name '*kernel_stats.csv' | head -1); cp $f $O/${name}_kernel_stats.csv; rm -rf $O/$name
  grep '"metric"' $O/$name.bench.log > $O/${name}_bench_line.json
}
pmc() { # name, counters, bench args
  name=$1; cnt=$2; shift; shift
  rocprofv3 --kernel-trace --pmc $cnt --output-format csv -d $O/$name -o run -- python $R/bench.py "$@" --single-run > $O/$name.out 2> $O/$name.err
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
}
stats c3 --workload c3 --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads
pmc pmc_c3_fetch FETCH_SIZE --workload c3
pmc pmc_c3_write WRITE_SIZE --workload c3
pmc pmc_c3_sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" --workload c3
stats c3drift --workload c3drift --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads
pmc pmc_c3drift_fetch FETCH_SIZE --workload c3drift
pmc pmc_c3drift_write WRITE_SIZE --workload c3drift
stats c4shard --workload c4shard --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads
ls $O | head -60
