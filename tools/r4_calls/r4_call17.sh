# Round 4, GPU call 17: k_advance_run with a lane per EVENT in pass 2 (the wave's rows dealt to its lanes) instead of a lane walking
# its user's run: the parity tests that reach it; C3 with drift at caps 8 / 16 / 32 / 64 events per round, the kernel compiled for
# 2 (default) / 3 / 4 waves per SIMD (-DRG_ADV_RUN_WAVES, builds of this call only); C5 at 32 / 64.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "run_ahead or lock_step_to_the_end or repacked or reproduces_reference_fixture or matches_oracle or repack_and_tail" > $O/gpu_tests17.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests17.log; tail -12 $O/gpu_tests17.log | cut -c1-600
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py $B $WL 2>$O/ab17_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', workload=d['config']['workload'].split(':')[0], events_per_s=round(d['value']/1e6,1), ms_per_step=round(d['ms_per_step'],2), kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" >> $O/ab_call17_run_ahead_lane_per_event.jsonl
}
rm -f $O/ab_call17_run_ahead_lane_per_event.jsonl
B="--steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --users 4000000"
WL="--workload c3drift"
run drift_lockstep RECOGYM_RUN_AHEAD=0
run drift_rounds8 RECOGYM_RUN_AHEAD=8
run drift_rounds16 RECOGYM_RUN_AHEAD=16
run drift_rounds32 A=1
run drift_rounds64 RECOGYM_RUN_AHEAD=64
run drift_rounds32_3waves RECOGYM_HIP_LIB=$R/recogym_amd/csrc/librecogym_hip_adv3.so
run drift_rounds32_4waves RECOGYM_HIP_LIB=$R/recogym_amd/csrc/librecogym_hip_adv4.so
B="--steps 1 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
WL="--workload c5"
run c5_rounds32 A=1
run c5_rounds64 RECOGYM_RUN_AHEAD=64
cat $O/ab_call17_run_ahead_lane_per_event.jsonl
