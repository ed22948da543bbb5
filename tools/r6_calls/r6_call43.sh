# Round 6, GPU call 43: what bounds k_logreg_screen — counter passes (own runs) on `bench.py --workload c5 --single-run`, fp16 and 8-bit rows.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pmc() { # name, env, counters
  name=$1; envs=$2; cnt=$3
  env $envs timeout 900 rocprofv3 --kernel-trace --pmc $cnt --output-format csv -d $O/$name -o run -- python $R/bench.py --workload c5 --single-run --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads > $O/$name.out 2> $O/$name.err
  f=$(find $O/$name -name '*counter_collection.csv' | head -1)
  python - "$f" "$O/${name}_counters.csv" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    if 'logreg' not in k: continue
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if (k, r['Dispatch_Id']) not in seen:
        seen.add((k, r['Dispatch_Id'])); n[k] += 1; dur[k] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
with open(sys.argv[2], 'w') as f:
    f.write('kernel,dispatches,total_ns,counter,value\n')
    for k in sorted(agg, key=lambda k: -dur[k]):
        for c, v in sorted(agg[k].items()):
            f.write(f'"{k[:50]}",{n[k]},{dur[k]},{c},{v:.6g}\n')
PY
  rm -rf $O/$name $O/$name.out
  grep screen $O/${name}_counters.csv | cut -d, -f2-
}
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM"
B="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_PENDING_STALL_CYCLES_sum TCC_TAG_STALL_sum"
pmc pmc_call43_lr_fp16_sq "RECOGYM_LOGREG=fp16" "$A"
pmc pmc_call43_lr_fp16_tcc "RECOGYM_LOGREG=fp16" "$B"
pmc pmc_call43_lr_fp16_fetch "RECOGYM_LOGREG=fp16" "FETCH_SIZE"
pmc pmc_call43_lr_int8_sq "RECOGYM_LOGREG=int8" "$A"
pmc pmc_call43_lr_int8_tcc "RECOGYM_LOGREG=int8" "$B"
