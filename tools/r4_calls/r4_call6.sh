# Round 4, GPU call 6: event-kind bias of k_walk2 below the old default; the per-user gym path with an arbitrary Python agent
# (rg_sim_step_user: one read-back per event); the sampled-oracle parity check at FULL bench sizes (profiles/r4 artefact).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 900 python tools/pipe_probe.py --reps 3 \
  bias8:RECOGYM_WALK_BIAS=8 bias4:RECOGYM_WALK_BIAS=4 bias3:RECOGYM_WALK_BIAS=3 bias2:RECOGYM_WALK_BIAS=2 bias1:RECOGYM_WALK_BIAS=1 bias0:RECOGYM_WALK_BIAS=0 \
  bias4s24:RECOGYM_WALK_BIAS=4,RECOGYM_WALK_SEARCH_BATCH=24 bias3s24:RECOGYM_WALK_BIAS=3,RECOGYM_WALK_SEARCH_BATCH=24 \
  > $O/ab_call6_walk_bias.jsonl 2> $O/ab_call6.err; echo "probe rc=$?"; tail -3 $O/ab_call6.err
python - <<'PY'
import json
for l in open('gpurun_out/r4/ab_call6_walk_bias.jsonl'):
    d = json.loads(l)
    print(d['config'], d['best_ms'], d['run_ms'], {k: d['profile'][k] for k in ('draw_exact_ms','walk1_ms','walk2_ms')})
PY
timeout 600 python tools/per_user_path.py --users 40 > $O/per_user_path.jsonl 2> $O/per_user_path.err; echo "per-user rc=$?"; cat $O/per_user_path.jsonl | cut -c1-400
rm -f $O/oracle_spot_check_full_size.jsonl
timeout 1800 python tests/oracle_spot_check.py c3 c2 c3drift c4shard c5 c5trained --sample 2000 --out $O/oracle_spot_check_full_size.jsonl > $O/spot.log 2> $O/spot.err; echo "spot rc=$?"; tail -2 $O/spot.err
python - <<'PY'
import json
for l in open('gpurun_out/r4/oracle_spot_check_full_size.jsonl'):
    d = json.loads(l); print(d['workload'], d['arm'], d['users'], d['p_click_exported'], d['sampled_users'], d['rows_compared'], d['kinds'], d['oracle_seconds'])
PY
