# Round 6, GPU call 4: 16 counter shards per tile (one address sustains ~6 M atomics/s), omega32 inside the 128-byte record — parity tests that reach it, c3drift / C5 with and without it.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "lds_search or (certificate_is_sound_for_uniforms and f16) or fused_and_sliced or run_ahead_rounds" 2>&1 | tail -15 > $O/gpu_tests_call4.txt
run() { # name, env, args
  name=$1; envs=$2; shift; shift
  env $envs timeout 300 python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab1.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events=d['config']['events_per_step'], ctr=d['config']['ctr'], ms_per_step=round(d['ms_per_step'],2), value=d['value'], kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items()})))" | tee -a $O/ab_call4_tp.jsonl
}
rm -f $O/ab_call4_tp.jsonl
run c3drift_tp "RECOGYM_SWEEP_LDS=1" --workload c3drift
run c3drift_old "RECOGYM_SWEEP_LDS=0" --workload c3drift
run c5_tp "RECOGYM_SWEEP_LDS=1" --workload c5
run c5_old "RECOGYM_SWEEP_LDS=0" --workload c5
