# Round 4, GPU call 14: the sampled-oracle check at full bench sizes again with the FINAL code (the walk's history line, the sweep
# writing the finalize output, the sliced float64 batch and the options refactor came after call 6's artefact); C5's 10^4-class
# LogReg arm with 150 sampled users (its oracle replay costs 500 s per 2 000 users; that path did not change since call 6).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
rm -f $O/oracle_spot_check_full_size_final.jsonl
timeout 1200 python tests/oracle_spot_check.py c3 c2 c3drift c4shard c5trained --sample 2000 --out $O/oracle_spot_check_full_size_final.jsonl > $O/spot14.log 2> $O/spot14.err; echo "spot rc=$?"; tail -2 $O/spot14.err
timeout 900 python tests/oracle_spot_check.py c5 --sample 150 --out $O/oracle_spot_check_full_size_final.jsonl >> $O/spot14.log 2>> $O/spot14.err; echo "spot c5 rc=$?"
python - <<'PY'
import json
for l in open('gpurun_out/r4/oracle_spot_check_full_size_final.jsonl'):
    d = json.loads(l); print(d['workload'], d['arm'], d['users'], d['p_click_exported'], d['sampled_users'], d['rows_compared'], d['oracle_seconds'])
PY
