# Round 6, GPU call 61: the cap on the events a round takes a user through (RECOGYM_RUN_AHEAD, default 32) on c3drift and C5.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
run() { # name, env, args
  name=$1; envs=$2; shift; shift
  env $envs timeout 300 python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab61.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', ms_per_step=round(d['ms_per_step'],2), value=d['value'], advance=[v['ms'] for k,v in d['kernels'].items() if 'advance' in k])))" | tee -a $O/ab_call61_run_ahead.jsonl
}
rm -f $O/ab_call61_run_ahead.jsonl $O/ab61.err
for h in 32 8 16 64; do run c3drift_hops$h "RECOGYM_RUN_AHEAD=$h" --workload c3drift; done
for h in 32 12 64; do run c5_hops$h "RECOGYM_RUN_AHEAD=$h" --workload c5; done
