# Round 6, GPU call 47: k_tail's population threshold at P x K = 2e5 (c3drift, C5; default 4096): never / 256 / 1024 / 16384.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
run() { # name, env, args
  name=$1; envs=$2; shift; shift
  env $envs timeout 600 python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab47.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', ms_per_step=round(d['ms_per_step'],2), value=d['value'])))" | tee -a $O/ab_call47_tail.jsonl
}
rm -f $O/ab_call47_tail.jsonl $O/ab47.err
for tb in 4096 0 256 1024 16384; do
  run c3drift_tail$tb "RECOGYM_TAIL=$tb" --workload c3drift
done
for tb in 4096 0 256 1024 16384; do
  run c5_tail$tb "RECOGYM_TAIL=$tb" --workload c5
done
