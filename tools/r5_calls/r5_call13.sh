# Round 5, GPU call 13: k_sweep_xh storing the chunk prefixes of two tiles together (32 bytes per user and store).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "pipelined_walk or walk_certificate or every_K_class or product_counts_around or error_free_sweep or sum_cache_does_not" 2>&1 | tail -4 > $O/gpu_tests_call13.txt
rm -f $O/ab_call13_xh.jsonl
for v in 0 0; do timeout 90 python tools/xh_probe.py 2000000 paired_stores 2>>$O/ab13.err | tail -1 >> $O/ab_call13_xh.jsonl; done
timeout 120 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab13.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='c3_xh_paired_stores', ms_per_step=round(d['ms_per_step'],2), kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items()})))" > $O/ab_call13_c3.jsonl
