# Round 4, GPU call 3: k_walk2's compact history line in PREFIX form (31 products, the act as a count of words below a key) against
# the 15-product line on C3; walk / pipeline / env drop-in (rg_sim_step_user) / sampled-oracle / multi-rank tests.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q -k "(pipelined or last_round or memo_and or sum_cache_matches or walk_certificate or ouc_integer or multi_rank or sampled_oracle or env_dropin or hip_matches_oracle or fixture) and not c5trained" > $O/gpu_tests3.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests3.log; tail -15 $O/gpu_tests3.log | cut -c1-600
timeout 900 python tools/pipe_probe.py --digest --reps 3 \
  line15:RECOGYM_WALK_HIST=1 \
  line31prefix:A=1 \
  > $O/ab_call3_history_line_prefix.jsonl 2> $O/ab_call3.err; echo "probe rc=$?"; tail -3 $O/ab_call3.err
python - <<'PY'
import json
for l in open('gpurun_out/r4/ab_call3_history_line_prefix.jsonl'):
    d = json.loads(l)
    print(d['config'], d['best_ms'], d['run_ms'], {k: d['profile'][k] for k in ('draw_mfma_ms','draw_search_ms','draw_exact_ms','walk1_ms','walk2_ms')}, d.get('digest_equal_to_first'), d['counters']['exact_sweeps'])
PY
