# Round 4, GPU call 8: the prefix-form sweep leaving the finalize / prefix kernels' output itself (RECOGYM_FIN_IN_SWEEP=0: the two
# kernels over every user) on C3; walk / cache / pipeline tests.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -k "pipelined or last_round or memo_and or sum_cache or walk_certificate or wide_logit or user_major or every_K or product_counts" > $O/gpu_tests8.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests8.log; tail -6 $O/gpu_tests8.log | cut -c1-400
timeout 900 python tools/pipe_probe.py --digest --reps 3 \
  kernels:RECOGYM_FIN_IN_SWEEP=0 in_sweep:A=1 \
  > $O/ab_call8_finalize_in_sweep.jsonl 2> $O/ab_call8.err; echo "probe rc=$?"; tail -3 $O/ab_call8.err
python - <<'PY'
import json
for l in open('gpurun_out/r4/ab_call8_finalize_in_sweep.jsonl'):
    d = json.loads(l)
    print(d['config'], d['best_ms'], d['run_ms'], {k: d['profile'][k] for k in ('draw_mfma_ms','draw_search_ms','draw_exact_ms','walk1_ms','walk2_ms')}, d.get('digest_equal_to_first'), d['counters']['exact_sweeps'])
PY
