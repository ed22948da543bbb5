# Round 6, GPU call 21: the frozen-LogReg screening pass from an 8-bit copy of coef^T (a quarter of the fp32 row bytes) — every LogReg
# parity test, C5 with it and with the fp16 copy.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/ -q -m gpu -k "logreg or c5 or verify_agents or test_agent or external_actions" 2>&1 | tail -6 > $O/gpu_tests_call21.txt
run() { # name, env, args
  name=$1; envs=$2; shift; shift
  env $envs timeout 600 python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab21.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events=d['config']['events_per_step'], ctr=d['config']['ctr'], ms_per_step=round(d['ms_per_step'],2), value=d['value'], kernels={k:[v['ms'], v.get('units'), v.get('float64_refined_acts')] for k,v in d['kernels'].items()})))" | tee -a $O/ab_call21_c5.jsonl
}
rm -f $O/ab_call21_c5.jsonl
run c5_int8 "RECOGYM_LOGREG=int8" --workload c5
run c5_fp16 "RECOGYM_LOGREG=fp16" --workload c5
run c5trained_int8 "RECOGYM_LOGREG=int8" --workload c5trained
run c5trained_fp16 "RECOGYM_LOGREG=fp16" --workload c5trained
