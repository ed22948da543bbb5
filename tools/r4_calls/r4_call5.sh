# Round 4, GPU call 5: k_walk2's iteration parameters again, now that the act is a count on the compact line (the balance between
# the event kinds moved): event-kind bias, click batch, search batch, refill threshold, hand-over threshold.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 1500 python tools/pipe_probe.py --reps 2 \
  default:A=1 \
  bias4:RECOGYM_WALK_BIAS=4 bias6:RECOGYM_WALK_BIAS=6 bias12:RECOGYM_WALK_BIAS=12 bias16:RECOGYM_WALK_BIAS=16 \
  click4:RECOGYM_WALK_CLICK_BATCH=4 click16:RECOGYM_WALK_CLICK_BATCH=16 \
  search12:RECOGYM_WALK_SEARCH_BATCH=12 search24:RECOGYM_WALK_SEARCH_BATCH=24 \
  refill4:RECOGYM_WALK_REFILL=4 refill16:RECOGYM_WALK_REFILL=16 \
  hand16:RECOGYM_WALK_HANDOVER=16 hand48:RECOGYM_WALK_HANDOVER=48 \
  > $O/ab_call5_walk_parameters.jsonl 2> $O/ab_call5.err; echo "probe rc=$?"; tail -3 $O/ab_call5.err
python - <<'PY'
import json
for l in open('gpurun_out/r4/ab_call5_walk_parameters.jsonl'):
    d = json.loads(l)
    print(d['config'], d['best_ms'], d['run_ms'], {k: d['profile'][k] for k in ('draw_exact_ms','walk1_ms','walk2_ms')})
PY
