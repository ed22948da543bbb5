# Round 6, GPU call 19: the walk's hand-over threshold by shard size (call 14: 16 beats 32 at 1.25 M users).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
rm -f $O/ab_call19_handover.jsonl
run() { # name, env, users
  name=$1; envs=$2; users=$3
  env $envs timeout 300 python bench.py --workload c3 --users $users --steps 5 --warmup 2 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab19.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', users=$users, ms_per_step=round(d['ms_per_step'],2), value=round(d['value']/1e9,3), kernels={k:v['ms'] for k,v in d['kernels'].items()})))" | tee -a $O/ab_call19_handover.jsonl
}
for u in 1250000 2500000 5000000 10000000; do
  for ho in 0 4 8 16 32; do run handover$ho "RECOGYM_WALK_HANDOVER=$ho" $u; done
done
