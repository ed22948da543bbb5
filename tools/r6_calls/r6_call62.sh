# Round 6, GPU call 62: a longer soak of the determinism probe on the C4 shard (12 runs) and on c3drift (6 runs).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
rm -f $O/determinism_call62.jsonl
timeout 400 python tools/determinism_probe.py c4shard 12 0 2>/dev/null | tail -1 >> $O/determinism_call62.jsonl
timeout 400 python tools/determinism_probe.py c3drift 6 1 2>/dev/null | tail -1 >> $O/determinism_call62.jsonl
python - <<'PY'
import json
for l in open('/root/repo/gpurun_out/r6/determinism_call62.jsonl'):
    d = json.loads(l); print(d['workload'], d['runs'], 'runs; rows differing per run:', [x.get('rows_differing') for x in d['diffs']])
PY
