# Round 6, GPU call 34: k_logreg_screen items range-major (all waves in flight in one eighth of the table's columns) against act-major.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
run() { # name, env, args
  name=$1; envs=$2; shift; shift
  env $envs timeout 600 python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab34.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events=d['config']['events_per_step'], ctr=d['config']['ctr'], ms_per_step=round(d['ms_per_step'],2), value=d['value'], kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items() if 'logreg' in k})))" | tee -a $O/ab_call34_c5.jsonl
}
rm -f $O/ab_call34_c5.jsonl $O/ab34.err
L=$R/recogym_amd/csrc/librecogym_hip
run c5_act_major "X=1" --workload c5
run c5_range_major "RECOGYM_HIP_LIB=${L}_lrrm.so" --workload c5
run c5_range_major_occ5 "RECOGYM_HIP_LIB=${L}_lrrm5.so" --workload c5
run c5_range_major_int8 "RECOGYM_HIP_LIB=${L}_lrrm.so RECOGYM_LOGREG=int8" --workload c5
tail -3 $O/ab34.err
