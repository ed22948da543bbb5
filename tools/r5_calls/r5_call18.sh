# Round 5, GPU call 18: with helpers a bandit iteration is full whatever the number of bandit lanes — so the organic kinds can wait for
# more lanes.  C3: event-kind bias 4 / 3 / 2 / 1 (organic when n_org * bias >= n_bandit * 4), helpers 3 / 5 / 7, search batch 16 / 24 / 32.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
rm -f $O/ab_call18_walk_tuning.jsonl
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 150 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab18.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events=d['config'].get('events_per_step'), ms_per_step=round(d['ms_per_step'],2), kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items()})))" >> $O/ab_call18_walk_tuning.jsonl
}
run h3_bias4 RECOGYM_WALK_HELPERS=3
run h3_bias3 RECOGYM_WALK_HELPERS=3 RECOGYM_WALK_BIAS=3
run h3_bias2 RECOGYM_WALK_HELPERS=3 RECOGYM_WALK_BIAS=2
run h3_bias1 RECOGYM_WALK_HELPERS=3 RECOGYM_WALK_BIAS=1
run h7_bias4 RECOGYM_WALK_HELPERS=7
run h7_bias2 RECOGYM_WALK_HELPERS=7 RECOGYM_WALK_BIAS=2
run h7_bias1 RECOGYM_WALK_HELPERS=7 RECOGYM_WALK_BIAS=1
run h5_bias2 RECOGYM_WALK_HELPERS=5 RECOGYM_WALK_BIAS=2
run h7_bias2_s24 RECOGYM_WALK_HELPERS=7 RECOGYM_WALK_BIAS=2 RECOGYM_WALK_SEARCH_BATCH=24
run h7_bias2_s32 RECOGYM_WALK_HELPERS=7 RECOGYM_WALK_BIAS=2 RECOGYM_WALK_SEARCH_BATCH=32
run h7_bias2_c16 RECOGYM_WALK_HELPERS=7 RECOGYM_WALK_BIAS=2 RECOGYM_WALK_CLICK_BATCH=16
