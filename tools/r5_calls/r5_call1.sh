# Round 5, GPU call 1: k_sweep_xh (error-free leading accumulator) — the MFMA probe, the parity tests that reach it, C3 with / without it.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
timeout 60 tools/ubench/mfma_f16_exact.bin > $O/ubench_mfma_f16_exact.jsonl 2>&1; echo "ubench rc $?" >> $O/ubench_mfma_f16_exact.jsonl
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "pipelined_walk or walk_certificate or memo_and_anchored or every_K_class or sum_cache_matches or last_round_wave" 2>&1 | tail -15 > $O/gpu_tests_call1.txt
run() { # name, env, args
  name=$1; envs=$2; shift; shift
  env $envs timeout 120 python bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise 2>>$O/ab1.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events=d['config']['events_per_step'], ctr=d['config']['ctr'], ms_per_step=round(d['ms_per_step'],2), kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items()})))" | tee -a $O/ab_call1_xh.jsonl
}
rm -f $O/ab_call1_xh.jsonl
run c3_xh "RECOGYM_XH=1" --workload c3
run c3_old "RECOGYM_XH=0" --workload c3
run c2_xh "RECOGYM_XH=1" --workload c2
run c2_old "RECOGYM_XH=0" --workload c2
