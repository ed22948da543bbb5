# Round 4, GPU call 16: run-ahead rounds (k_advance_run: a user's whole bandit run per launch, per-user event indices) — the parity
# tests that reach it, then C3 with drift and C5 with an event per launch (RECOGYM_RUN_AHEAD=0) against rounds of <= 8 / 32 / 128 events.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "run_ahead or lock_step_to_the_end or repacked or reproduces_reference_fixture or matches_oracle or repack_and_tail" > $O/gpu_tests16.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests16.log; tail -12 $O/gpu_tests16.log | cut -c1-600
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py $B $WL 2>$O/ab16_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', workload=d['config']['workload'], events_per_s=round(d['value']/1e6,1), ms_per_step=round(d['ms_per_step'],2), kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" >> $O/ab_call16_run_ahead.jsonl
}
rm -f $O/ab_call16_run_ahead.jsonl
B="--steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --users 4000000"
WL="--workload c3drift"
run drift_lockstep RECOGYM_RUN_AHEAD=0
run drift_rounds32 A=1
run drift_rounds8 RECOGYM_RUN_AHEAD=8
run drift_rounds128 RECOGYM_RUN_AHEAD=128
B="--steps 1 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
WL="--workload c5"
run c5_lockstep RECOGYM_RUN_AHEAD=0
run c5_rounds32 A=1
cat $O/ab_call16_run_ahead.jsonl
