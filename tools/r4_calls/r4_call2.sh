# Round 4, GPU call 2: compact view-history line of k_walk2 (31 products per 128-byte LDS line) against the 15-product line on C3;
# the sampled-oracle parity check at bench size; the multi-rank bench line on one device; walk / pipeline tests.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -k "(pipelined or last_round or memo_and or sum_cache_matches or walk_certificate or ouc_integer or multi_rank or sampled_oracle) and not c5trained" > $O/gpu_tests2.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests2.log; tail -15 $O/gpu_tests2.log | cut -c1-600
timeout 900 python tools/pipe_probe.py --digest --reps 3 \
  line15:RECOGYM_PIPE=1,RECOGYM_WALK_HIST=1 \
  line31:RECOGYM_PIPE=1 \
  > $O/ab_call2_history_line.jsonl 2> $O/ab_call2.err; echo "probe rc=$?"; tail -3 $O/ab_call2.err
python - <<'PY'
import json
for l in open('gpurun_out/r4/ab_call2_history_line.jsonl'):
    d = json.loads(l)
    print(d['config'], d['best_ms'], d['run_ms'], {k: d['profile'][k] for k in ('draw_mfma_ms','draw_search_ms','draw_exact_ms','walk1_ms','walk2_ms')}, d.get('digest_equal_to_first'), d['counters']['exact_sweeps'])
PY
