# Round 5, GPU call 22: the click batch inside a bandit iteration (walk_click_join), wavefront-scope fences around the helpers' LDS
# tables.  Parity of the walked run, C3 / C2 with join 1 / 0, iterations by kind.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "walk or memo or sum_cache or hip_matches_oracle or organic_only or phantom" 2>&1 | tail -4 > $O/gpu_tests_call22.txt
rm -f $O/ab_call22_click_join.jsonl
run() {  # name, workload, env...
  name=$1; wl=$2; shift; shift
  env "$@" timeout 150 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab22.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events=d['config'].get('events_per_step'), ms_per_step=round(d['ms_per_step'],2), kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items()})))" >> $O/ab_call22_click_join.jsonl
}
run c3_join1 c3 RECOGYM_WALK_CLICK_JOIN=1
run c3_join0 c3 RECOGYM_WALK_CLICK_JOIN=0
run c3_join1_click4 c3 RECOGYM_WALK_CLICK_JOIN=1 RECOGYM_WALK_CLICK_BATCH=4
run c3_join1_click16 c3 RECOGYM_WALK_CLICK_JOIN=1 RECOGYM_WALK_CLICK_BATCH=16
run c3_join1 c3 RECOGYM_WALK_CLICK_JOIN=1
run c2_join1 c2 RECOGYM_WALK_CLICK_JOIN=1
run c2_join0 c2 RECOGYM_WALK_CLICK_JOIN=0
L=$R/recogym_amd/csrc/librecogym_hip_walktiming.so
RECOGYM_HIP_LIB=$L timeout 200 python tools/walk_kinds.py c3 2>>$O/ab22.err | tail -1 > $O/walk_kinds_call22.jsonl
