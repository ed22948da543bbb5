# Round 4, GPU call 1: the pipelined walk (run_walk_pipe) — oracle parity at small size, then the execution forms on C3 at full size
# (10 M users): run times, per-kernel HIP-event times, counters and the log checksum of every form.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "pipelined or last_round or memo_and or sum_cache_matches" > $O/gpu_tests1.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests1.log; tail -5 $O/gpu_tests1.log | cut -c1-400
timeout 1500 python tools/pipe_probe.py --digest --reps 3 \
  serial:RECOGYM_PIPE=0 \
  g1:RECOGYM_PIPE=1 \
  g4m0:RECOGYM_PIPE=4,RECOGYM_PIPE_MODE=0 \
  g4m1:RECOGYM_PIPE=4,RECOGYM_PIPE_MODE=1 \
  g4m1o2:RECOGYM_PIPE=4,RECOGYM_PIPE_MODE=1,RECOGYM_PIPE_OCC1=2 \
  g4m1o2x256:RECOGYM_PIPE=4,RECOGYM_PIPE_MODE=1,RECOGYM_PIPE_OCC1=2,RECOGYM_PIPE_XBLOCKS=256 \
  g4m2:RECOGYM_PIPE=4,RECOGYM_PIPE_MODE=2 \
  g4m2o2:RECOGYM_PIPE=4,RECOGYM_PIPE_MODE=2,RECOGYM_PIPE_OCC1=2 \
  g8m1:RECOGYM_PIPE=8,RECOGYM_PIPE_MODE=1 \
  g8m2:RECOGYM_PIPE=8,RECOGYM_PIPE_MODE=2 \
  g2m1:RECOGYM_PIPE=2,RECOGYM_PIPE_MODE=1 \
  > $O/ab_call1_pipe_forms.jsonl 2> $O/ab_call1.err; echo "probe rc=$?"; tail -3 $O/ab_call1.err
python - <<'PY'
import json
for l in open('gpurun_out/r4/ab_call1_pipe_forms.jsonl'):
    d = json.loads(l)
    print(d['config'], d['best_ms'], d['run_ms'], {k: d['profile'][k] for k in ('draw_mfma_ms','draw_search_ms','draw_exact_ms','walk1_ms','walk2_ms')}, d.get('digest_equal_to_first'), d['counters']['exact_sweeps'])
PY
