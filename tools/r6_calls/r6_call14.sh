# Round 6, GPU call 14: a rank's share of a strongly scaled C3 (1.25 M / 2.5 M users) — the walk's hand-over threshold, the
# float64 batch's grid and its matrix / vector mix at shard size.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
rm -f $O/ab_call14_shard_tuning.jsonl
run() { # name, env, users
  name=$1; envs=$2; users=$3
  env $envs timeout 200 python bench.py --workload c3 --users $users --steps 5 --warmup 2 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab14.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', users=$users, ms_per_step=round(d['ms_per_step'],2), value=round(d['value']/1e9,3), kernels={k:v['ms'] for k,v in d['kernels'].items()})))" | tee -a $O/ab_call14_shard_tuning.jsonl
}
run default "X=1" 1250000
run handover16 "RECOGYM_WALK_HANDOVER=16" 1250000
run handover48 "RECOGYM_WALK_HANDOVER=48" 1250000
run handover64 "RECOGYM_WALK_HANDOVER=64" 1250000
run xblocks256 "RECOGYM_PIPE_XBLOCKS=256" 1250000
run xblocks512 "RECOGYM_PIPE_XBLOCKS=512" 1250000
run mix4 "RECOGYM_EXACT_MIX=4" 1250000
run mix7 "RECOGYM_EXACT_MIX=7" 1250000
run refill4 "RECOGYM_WALK_REFILL=4" 1250000
run refill16 "RECOGYM_WALK_REFILL=16" 1250000
run default "X=1" 2500000
run handover48 "RECOGYM_WALK_HANDOVER=48" 2500000
run handover64 "RECOGYM_WALK_HANDOVER=64" 2500000
