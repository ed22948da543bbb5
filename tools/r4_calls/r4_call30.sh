# Round 4, GPU call 30: the one-set sweep at two blocks per CU on C3; the one-set three-block build on C3 with drift (2 M users).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
run() { # name, lib, args
  name=$1; lib=$2; shift; shift
  RECOGYM_HIP_LIB=$lib timeout 25 python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise 2>>$O/ab30.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events=d['config']['events_per_step'], ctr=d['config']['ctr'], ms_per_step=round(d['ms_per_step'],2), kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" | tee -a $O/ab_call30_sweep_one_set.jsonl
}
rm -f $O/ab_call30_sweep_one_set.jsonl
run c3_one_set_two_blocks $R/recogym_amd/csrc/librecogym_hip_oneset2.so --workload c3
run drift2m_one_set_three_blocks $R/recogym_amd/csrc/librecogym_hip_oneset3.so --workload c3drift --users 2000000
run drift2m_default $R/recogym_amd/csrc/librecogym_hip.so --workload c3drift --users 2000000
