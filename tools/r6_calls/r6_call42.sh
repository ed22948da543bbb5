# Round 6, GPU call 42: the fp16 LogReg screen at 20 classes per lane for class ranges > 512 (C5): parity (incl. 4 168 classes: the shifted last lane), C5 fp16 / 8-bit, c5trained.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_hip_parity.py -q -m gpu -k "logreg" 2>&1 | tail -5 > $O/gpu_tests_call42.txt
cat $O/gpu_tests_call42.txt
run() { # name, env, args
  name=$1; envs=$2; shift; shift
  env $envs timeout 600 python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab42.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events=d['config']['events_per_step'], ctr=d['config']['ctr'], ms_per_step=round(d['ms_per_step'],2), value=d['value'], kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items() if 'logreg' in k})))" | tee -a $O/ab_call42_c5.jsonl
}
rm -f $O/ab_call42_c5.jsonl $O/ab42.err
run c5_fp16 "X=1" --workload c5
run c5_int8_20_per_lane "RECOGYM_LOGREG=int8" --workload c5
run c5_fp16_b "X=1" --workload c5
run c5_int8_20_per_lane_b "RECOGYM_LOGREG=int8" --workload c5
run c5trained "X=1" --workload c5trained
tail -3 $O/ab42.err
