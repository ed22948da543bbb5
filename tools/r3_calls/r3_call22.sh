# Round 3, GPU call 22: walk parameters with the final k_walk2 (click batch, bias, hand-over, refill).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
run() { # name, env...
  name=$1; shift
  env "$@" timeout 60 python bench.py $B $WL 2>$O/ab22_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(json.dumps(dict(name='$name', ms_per_step=round(d['ms_per_step'],2), round1_ms=r.get('round1_ms'), later_rounds_ms=r.get('later_rounds_ms'), walk=d['kernels']['walk']['ms'])))" >> $O/ab22.jsonl
}
rm -f $O/ab22.jsonl
B="--steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
WL="--workload c3"
run default A=1
run click12 RECOGYM_WALK_CLICK_BATCH=12
run click6 RECOGYM_WALK_CLICK_BATCH=6
run bias0 RECOGYM_WALK_BIAS=0
run bias4 RECOGYM_WALK_BIAS=4
run bias16 RECOGYM_WALK_BIAS=16
run handover16 RECOGYM_WALK_HANDOVER=16
run handover48 RECOGYM_WALK_HANDOVER=48
run refill4 RECOGYM_WALK_REFILL=4
run refill16 RECOGYM_WALK_REFILL=16
cat $O/ab22.jsonl
