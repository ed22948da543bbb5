# Round 4, GPU call 4: the whole GPU suite (pipeline, compact history line, prefix search of the lock-step sweep, sampled-oracle
# parity incl. the reference-fitted c5 policies, rg_sim_step_user); the lock-step sweep's search on prefixes against the float64
# running sums on C3 with drift and C5.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests4.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests4.log; tail -12 $O/gpu_tests4.log | cut -c1-600
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py $B $WL 2>$O/ab4_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events_per_s=round(d['value']/1e6,1), ms_per_step=round(d['ms_per_step'],2), kernels={k:(v['ms']) for k,v in d['kernels'].items()}, exact_fraction=d['roofline'].get('exact_fraction'))))" >> $O/ab_call4_lock_prefix.jsonl
}
rm -f $O/ab_call4_lock_prefix.jsonl
B="--steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
WL="--workload c3drift"
run c3drift_float64_sums RECOGYM_LOCK_PREFIX=0
run c3drift_prefix A=1
B="--steps 1 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
WL="--workload c5"
run c5_float64_sums RECOGYM_LOCK_PREFIX=0
run c5_prefix A=1
cat $O/ab_call4_lock_prefix.jsonl
