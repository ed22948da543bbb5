# Round 5, GPU call 10: k_sweep_xh with the super-chunk prefixes staged in LDS (one row per user at the end) and the {sum, reference}
# records only for users whose reference moves: parity (incl. wide logit ranges through the walked run), the sweep's time, C3.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "pipelined_walk or walk_certificate or memo_and_anchored or every_K_class or product_counts_around or error_free_sweep or sum_cache_matches" 2>&1 | tail -6 > $O/gpu_tests_call10.txt
rm -f $O/ab_call10_xh.jsonl
for v in 0 0; do
  RECOGYM_HIP_LIB=$R/recogym_amd/csrc/librecogym_hip.so timeout 90 python tools/xh_probe.py 2000000 staged_scp 2>>$O/ab10.err | tail -1 >> $O/ab_call10_xh.jsonl
done
timeout 120 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab10.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='c3_xh_staged_scp', ms_per_step=round(d['ms_per_step'],2), kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items()})))" > $O/ab_call10_c3.jsonl
