# Round 4, GPU call 22: where k_advance_run's time goes — timing build (-DRG_ADV_TIMING, results wrong by construction): without
# pass 2, without the act, without the row stores; rounds of one event (RECOGYM_RUN_AHEAD=1).  C3 with drift at 4 M users, C5.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
X=$R/recogym_amd/csrc/librecogym_hip_advtiming.so
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py $B $WL 2>$O/ab22_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', workload=d['config']['workload'].split(':')[0], events_per_s=round(d['value']/1e6,1), ms_per_step=round(d['ms_per_step'],2), kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" >> $O/ab_call22_advance_run_timing.jsonl
}
rm -f $O/ab_call22_advance_run_timing.jsonl
B="--steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
WL="--workload c3drift --users 4000000"
run drift_full RECOGYM_HIP_LIB=$X
run drift_no_pass2 RECOGYM_HIP_LIB=$X RECOGYM_ABLATE=16777216
run drift_no_act RECOGYM_HIP_LIB=$X RECOGYM_ABLATE=33554432
run drift_no_stores RECOGYM_HIP_LIB=$X RECOGYM_ABLATE=67108864
run drift_no_act_no_stores RECOGYM_HIP_LIB=$X RECOGYM_ABLATE=100663296
run drift_one_event_rounds RECOGYM_HIP_LIB=$X RECOGYM_RUN_AHEAD=1
WL="--workload c5"
run c5_full RECOGYM_HIP_LIB=$X
run c5_no_pass2 RECOGYM_HIP_LIB=$X RECOGYM_ABLATE=16777216
cat $O/ab_call22_advance_run_timing.jsonl
