# Round 5, GPU call 19: where k_walk2's time goes now that the bandit iterations are full (-DRG_WALK_TIMING build, RECOGYM_ABLATE
# bits 23..29: results wrong by design).  C3, new defaults (helpers 7, bias 2, search batch 24).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
rm -f $O/ab_call19_walk_timing.jsonl
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 150 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab19.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events=d['config'].get('events_per_step'), ms_per_step=round(d['ms_per_step'],2), kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items()})))" >> $O/ab_call19_walk_timing.jsonl
}
L=$R/recogym_amd/csrc/librecogym_hip_walktiming.so
run default
run timing_build RECOGYM_HIP_LIB=$L
run all_memo_hits_no_row_read RECOGYM_HIP_LIB=$L RECOGYM_ABLATE=$((1<<23))
run no_view_insertion RECOGYM_HIP_LIB=$L RECOGYM_ABLATE=$((1<<27))
run no_policy_act RECOGYM_HIP_LIB=$L RECOGYM_ABLATE=$((1<<28))
run no_row_store RECOGYM_HIP_LIB=$L RECOGYM_ABLATE=$((1<<25))
run no_beta_row RECOGYM_HIP_LIB=$L RECOGYM_ABLATE=$((1<<24))
run memo_hits_and_no_view RECOGYM_HIP_LIB=$L RECOGYM_ABLATE=$(((1<<23)|(1<<27)))
run memo_hits_no_view_no_act_no_row RECOGYM_HIP_LIB=$L RECOGYM_ABLATE=$(((1<<23)|(1<<27)|(1<<28)|(1<<25)|(1<<24)))
run helpers0 RECOGYM_WALK_HELPERS=0
