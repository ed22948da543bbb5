# Round 5, GPU call 17: k_walk2 with HELPERS — the lanes that sit a bandit iteration out take the next events of the bandit runs of
# the lanes that are in it (walk_helpers = 0 .. 3 events per owner).  Parity of the walked run first, then C3 / C2 with 0, 1, 3.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "walk or memo or sum_cache or hip_matches_oracle or organic_only or phantom" 2>&1 | tail -8 > $O/gpu_tests_call17.txt
rm -f $O/ab_call17_helpers.jsonl
for h in 0 1 3 0 3; do
  RECOGYM_WALK_HELPERS=$h timeout 150 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab17.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='c3_helpers_$h', events=d['config'].get('events_per_step'), ms_per_step=round(d['ms_per_step'],2), kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items()})))" >> $O/ab_call17_helpers.jsonl
done
for h in 0 3; do
  RECOGYM_WALK_HELPERS=$h timeout 100 python bench.py --workload c2 --steps 5 --warmup 2 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab17.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='c2_helpers_$h', events=d['config'].get('events_per_step'), ms_per_step=round(d['ms_per_step'],2), kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items()})))" >> $O/ab_call17_helpers.jsonl
done
