# Round 4, GPU call 21: k_advance_run — pass 1 counted by the whole wave (idle lanes work ahead for the users still counting) and
# pass 2 without the event draw of events that cannot click: parity tests that reach the rounds, then C3 with drift (4 M users),
# C5, C5 with fitted policies.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "run_ahead or lock_step_to_the_end or repacked or reproduces_reference_fixture or matches_oracle or repack_and_tail" > $O/gpu_tests21.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests21.log; tail -5 $O/gpu_tests21.log | cut -c1-600
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py $B $WL 2>$O/ab21_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', workload=d['config']['workload'].split(':')[0], events_per_s=round(d['value']/1e6,1), ms_per_step=round(d['ms_per_step'],2), kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" >> $O/ab_call21_advance_run_wave_count.jsonl
}
rm -f $O/ab_call21_advance_run_wave_count.jsonl
B="--steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
WL="--workload c3drift --users 4000000"; run c3drift_4m A=1
WL="--workload c5"; run c5 A=1
WL="--workload c5trained"; run c5trained A=1
cat $O/ab_call21_advance_run_wave_count.jsonl
