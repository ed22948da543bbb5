# Round 6, GPU call 48: no k_tail beyond P x K = 10^6 — the oracle cases (incl. P = 10^5, K = 64), the forms test, then the C4 shard line.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "matches_the_oracle and not logreg" 2>&1 | tail -4 > $O/gpu_tests_call48.txt
cat $O/gpu_tests_call48.txt
timeout 600 python bench.py --workload c4shard --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='c4shard_no_tail', ms_per_step=round(d['ms_per_step'],2), value=d['value'], kernels={k:v['ms'] for k,v in d['kernels'].items()})))" | tee $O/ab_call48_c4.jsonl
