# Round 6, GPU call 53: k_draw_tpw's tile barrier before the first slot that touches the next tile: the C4 shard 6 times (every ordered log
# against the first), the wide-K parity tests, then C3 / C2 / C5's LogReg arm through the same probe.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
rm -f $O/determinism_call53.jsonl
timeout 300 python tools/determinism_probe.py c4shard 6 1 2>/dev/null | tail -1 >> $O/determinism_call53.jsonl
timeout 300 python tools/determinism_probe.py c4shard 6 0 2>/dev/null | tail -1 >> $O/determinism_call53.jsonl
cut -c1-400 $O/determinism_call53.jsonl
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "every_K_class or lds_search or (matches_the_oracle and not logreg)" 2>&1 | tail -3 > $O/gpu_tests_call53.txt
cat $O/gpu_tests_call53.txt
timeout 300 python tools/determinism_probe.py c3 4 0 2>/dev/null | tail -1 >> $O/determinism_call53.jsonl
timeout 300 python tools/determinism_probe.py c2 4 0 2>/dev/null | tail -1 >> $O/determinism_call53.jsonl
tail -2 $O/determinism_call53.jsonl | cut -c1-400
timeout 300 python bench.py --workload c4shard --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='c4shard', ms_per_step=round(d['ms_per_step'],2), value=d['value'], kernels={k:v['ms'] for k,v in d['kernels'].items()})))" | tee $O/ab_call53_c4.jsonl
