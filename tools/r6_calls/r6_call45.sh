# Round 6, GPU call 45: the population at which a C4 shard's run goes to k_tail (default 128 at P x K = 6.4e6): 32 / 64 / 256 / 512.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
run() { # name, env, args
  name=$1; envs=$2; shift; shift
  env $envs timeout 600 python bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab46.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', ms_per_step=round(d['ms_per_step'],2), value=d['value'], kernels={k:v['ms'] for k,v in d['kernels'].items()})))" | tee -a $O/ab_call46_c4_tail.jsonl
}
rm -f $O/ab_call46_c4_tail.jsonl $O/ab46.err
run c4_tail128 "X=1" --workload c4shard
run c4_tail32 "RECOGYM_TAIL=32" --workload c4shard
run c4_tail64 "RECOGYM_TAIL=64" --workload c4shard
run c4_tail256 "RECOGYM_TAIL=256" --workload c4shard
run c4_tail512 "RECOGYM_TAIL=512" --workload c4shard
