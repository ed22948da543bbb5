# Round 6, GPU call 29: k_logreg_screen<Q8> (fp16 / 8-bit + second level, the same short chain of round trips, no prefetch):
# parity, then C5 at 5 (default) / 4 / 6 waves per SIMD, fp16 and 8-bit rows.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_hip_parity.py -q -m gpu -k "logreg" 2>&1 | tail -5 > $O/gpu_tests_call29.txt
cat $O/gpu_tests_call29.txt
run() { # name, env, args
  name=$1; envs=$2; shift; shift
  env $envs timeout 600 python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab29.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events=d['config']['events_per_step'], ctr=d['config']['ctr'], ms_per_step=round(d['ms_per_step'],2), value=d['value'], kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items() if 'logreg' in k})))" | tee -a $O/ab_call29_c5.jsonl
}
rm -f $O/ab_call29_c5.jsonl $O/ab29.err
L=$R/recogym_amd/csrc/librecogym_hip
run c5_fp16_occ5 "X=1" --workload c5
run c5_int8_occ5 "RECOGYM_LOGREG=int8" --workload c5
run c5_fp16_occ4 "RECOGYM_HIP_LIB=${L}_lrocc4.so" --workload c5
run c5_int8_occ4 "RECOGYM_HIP_LIB=${L}_lrocc4.so RECOGYM_LOGREG=int8" --workload c5
run c5_fp16_occ6 "RECOGYM_HIP_LIB=${L}_lrocc6.so" --workload c5
run c5_int8_occ6 "RECOGYM_HIP_LIB=${L}_lrocc6.so RECOGYM_LOGREG=int8" --workload c5
run c5trained "X=1" --workload c5trained
tail -3 $O/ab29.err
