# Round 3, GPU call 3: k_walk2 at 4 blocks/CU (omega32 from memory, float64 pick as a call) vs 3 blocks/CU; SQ counters per
# walk dispatch; frozen LogReg: fp16 screen-and-refine vs fp32 scores on config 5.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q --maxfail=12 > $O/gpu_tests3.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests3.log; tail -30 $O/gpu_tests3.log | cut -c1-300
B="--steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py $B $WL 2>$O/ab3_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events_per_s=d['value'], ms_per_step=d['ms_per_step'], kernels={k:(v['ms']) for k,v in d['kernels'].items()}, walk=d['kernels'].get('walk'), roofline=d['roofline'])))" >> $O/ab3.jsonl
}
rm -f $O/ab3.jsonl
WL="--workload c3"
run c3_w2_occ3 RECOGYM_WALK_OCC=3
run c3_w2_occ4 RECOGYM_WALK_OCC=4
WL="--workload c2"
run c2_w2_occ3 RECOGYM_WALK_OCC=3
run c2_w2_occ4 RECOGYM_WALK_OCC=4
B="--steps 1 --warmup 1 --no-cpu-baseline"
WL="--workload c5"
run c5_fp16 RECOGYM_LOGREG=fp16
run c5_fp32 RECOGYM_LOGREG=fp32
cut -c1-330 $O/ab3.jsonl
# ---- SQ counters per walk dispatch (2 M users, one run per pass) ----
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_counter_names.txt
pmc() { # name, counters, env...
  name=$1; cnt=$2; shift; shift
  env "$@" RECOGYM_DEBUG=1 timeout 300 rocprofv3 --kernel-trace --pmc $cnt --output-format csv -d $O/pmc_$name -- python $R/bench.py --workload c3 --users 2000000 --single-run > $O/pmc_$name.out 2> $O/pmc_$name.err
  f=$(find $O/pmc_$name -name '*counter_collection.csv' | head -1)
  python - "$f" <<'PY' > $O/pmc_$name.txt
import csv, sys, collections
rows = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_walk' in r['Kernel_Name'] or 'k_cache_prefix' in r['Kernel_Name']:
        k = (int(r['Dispatch_Id']), r['Kernel_Name'][:40])
        rows[k][r['Counter_Name']] = float(r['Counter_Value'])
        rows[k]['ns'] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for k in sorted(rows):
    print(k, {a: f'{b:.4g}' for a, b in rows[k].items()})
PY
  grep "walk round" $O/pmc_$name.err | tail -2 >> $O/pmc_$name.txt
  rm -rf $O/pmc_$name
  cat $O/pmc_$name.txt | cut -c1-420
}
pmc w2_occ3_a "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" RECOGYM_WALK_OCC=3
pmc w2_occ4_a "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" RECOGYM_WALK_OCC=4
pmc w2_occ3_b "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES" RECOGYM_WALK_OCC=3
