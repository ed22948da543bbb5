# Round 6, GPU call 27: the LogReg screen with its rows in flight (no branch per history row: the compiler had sunk each row load
# into it — a dependent load pair and a full wait per row), fp16 and 8-bit + second level: parity, then C5 with both.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_hip_parity.py -q -m gpu -k "logreg" 2>&1 | tail -5 > $O/gpu_tests_call27.txt
cat $O/gpu_tests_call27.txt
run() { # name, env, args
  name=$1; envs=$2; shift; shift
  env $envs timeout 600 python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab27.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events=d['config']['events_per_step'], ctr=d['config']['ctr'], ms_per_step=round(d['ms_per_step'],2), value=d['value'], kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items()})))" | tee -a $O/ab_call27_c5.jsonl
}
rm -f $O/ab_call27_c5.jsonl $O/ab27.err
run c5_fp16 "RECOGYM_LOGREG=fp16" --workload c5
run c5_int8_two_level "RECOGYM_LOGREG=int8" --workload c5
run c5_fp16_b "RECOGYM_LOGREG=fp16" --workload c5
run c5_int8_two_level_b "RECOGYM_LOGREG=int8" --workload c5
tail -5 $O/ab27.err
