# Round 5, GPU call 20: the other walk parameters once more, with helpers (waiting lanes now work as helpers: larger batches are cheap).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
rm -f $O/ab_call20_walk_tuning2.jsonl
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 150 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab20.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events=d['config'].get('events_per_step'), ms_per_step=round(d['ms_per_step'],2), kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items()})))" >> $O/ab_call20_walk_tuning2.jsonl
}
run default
run bias0 RECOGYM_WALK_BIAS=0
run click24 RECOGYM_WALK_CLICK_BATCH=24
run click4 RECOGYM_WALK_CLICK_BATCH=4
run refill16 RECOGYM_WALK_REFILL=16
run refill4 RECOGYM_WALK_REFILL=4
run handover16 RECOGYM_WALK_HANDOVER=16
run handover48 RECOGYM_WALK_HANDOVER=48
run search20 RECOGYM_WALK_SEARCH_BATCH=20
run search28 RECOGYM_WALK_SEARCH_BATCH=28
run default2
