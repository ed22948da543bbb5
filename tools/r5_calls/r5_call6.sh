# Round 5, GPU call 6: k_sweep_xh with its stores issued right behind the tile barrier (a tile's time before the next vmcnt wait).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "pipelined_walk or walk_certificate or memo_and_anchored or every_K_class or product_counts_around or wide_logit_range" 2>&1 | tail -5 > $O/gpu_tests_call6.txt
rm -f $O/ab_call6_xh.jsonl
for v in 0 128 0; do
  lib=$R/recogym_amd/csrc/librecogym_hip_xhabl$v.so
  [ $v = 0 ] && lib=$R/recogym_amd/csrc/librecogym_hip.so
  RECOGYM_HIP_LIB=$lib timeout 90 python tools/xh_probe.py 2000000 abl$v 2>>$O/ab6.err | tail -1 >> $O/ab_call6_xh.jsonl
done
timeout 120 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab6.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='c3_xh_stores_behind_barrier', ms_per_step=round(d['ms_per_step'],2), kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items()})))" > $O/ab_call6_c3.jsonl
