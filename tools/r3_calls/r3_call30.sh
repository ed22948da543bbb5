# Round 3, GPU call 30: k_walk2 — lanes that missed the memo at which a wave runs the search (RECOGYM_WALK_SEARCH_BATCH, default 16).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
run() { # name, env...
  name=$1; shift
  env "$@" timeout 60 python bench.py $B $WL 2>$O/ab30_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(json.dumps(dict(name='$name', ms_per_step=round(d['ms_per_step'],2), round1_ms=r.get('round1_ms'), later_rounds_ms=r.get('later_rounds_ms'), walk=d['kernels']['walk']['ms'])))" >> $O/ab30.jsonl
}
rm -f $O/ab30.jsonl
B="--steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
WL="--workload c3"
for b in 16 8 12 24 32; do run search_batch_$b RECOGYM_WALK_SEARCH_BATCH=$b; done
cat $O/ab30.jsonl
