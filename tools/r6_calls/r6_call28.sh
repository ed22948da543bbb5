# Round 6, GPU call 28: k_logreg_screen rebuilt as a short chain of round trips (history header + first 8 entries in one, rows of a
# batch in one, the next batch's rows prefetched, v_fma_mix_f32 on the fp16 halves): parity, then C5 with 3 / 4 waves per SIMD
# (prefetch) and 5 / 4 (no prefetch).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_hip_parity.py -q -m gpu -k "logreg" 2>&1 | tail -5 > $O/gpu_tests_call28.txt
cat $O/gpu_tests_call28.txt
run() { # name, env, args
  name=$1; envs=$2; shift; shift
  env $envs timeout 600 python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab28.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events=d['config']['events_per_step'], ctr=d['config']['ctr'], ms_per_step=round(d['ms_per_step'],2), value=d['value'], kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items() if 'logreg' in k})))" | tee -a $O/ab_call28_c5.jsonl
}
rm -f $O/ab_call28_c5.jsonl $O/ab28.err
L=$R/recogym_amd/csrc/librecogym_hip
run c5_occ3_prefetch "X=1" --workload c5
run c5_occ4_prefetch "RECOGYM_HIP_LIB=${L}_lrocc4.so" --workload c5
run c5_occ5_noprefetch "RECOGYM_HIP_LIB=${L}_lrocc5np.so" --workload c5
run c5_occ4_noprefetch "RECOGYM_HIP_LIB=${L}_lrocc4np.so" --workload c5
run c5_int8 "RECOGYM_LOGREG=int8" --workload c5
run c5trained "X=1" --workload c5trained
tail -3 $O/ab28.err
