# Round 3, GPU call 16: k_walk2 with the memo-answered draws and the bandit events of a wave in ONE iteration (RECOGYM_WALK_BIAS=0).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
RECOGYM_WALK_BIAS=0 timeout 900 python -m pytest tests -m gpu -q -x -k "sum_cache or memo or wave_per_user or sigma_omega_zero or walk_certificate" > $O/gpu_tests16.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests16.log; tail -4 $O/gpu_tests16.log | cut -c1-400
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py $B $WL 2>$O/ab16_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(json.dumps(dict(name='$name', events_per_s=round(d['value']/1e6,1), ms_per_step=round(d['ms_per_step'],2), round1_ms=r.get('round1_ms'), later_rounds_ms=r.get('later_rounds_ms'), kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" >> $O/ab16.jsonl
}
rm -f $O/ab16.jsonl
B="--steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
for WLN in c3 c2; do WL="--workload $WLN"; run ${WLN}_default A=1; run ${WLN}_merged RECOGYM_WALK_BIAS=0; done
WL="--workload c3 --users 1250000"; run c3_shard_default A=1; run c3_shard_merged RECOGYM_WALK_BIAS=0
cat $O/ab16.jsonl
