# Round 6, GPU call 18: k_draw_tpw compiled with the VGPR form of the MFMA (no v_accvgpr_read in a one-wave stream bound by its
# issue rate), counters instead of integer divisions — parity, cycles per tile, step 0, the C4 shard.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_hip_parity.py -q -m gpu -k "lds_search or every_K_class or (hip_matches_oracle and (14 or 3 or 4))" 2>&1 | tail -4 > $O/gpu_tests_call18.txt
rm -f $O/ab_call18_wide_step0.txt
echo "clock probe build" >> $O/ab_call18_wide_step0.txt
RECOGYM_HIP_LIB=$R/recogym_amd/csrc/librecogym_hip_tpwclk.so timeout 200 python tools/wide_step0.py 500000 2>&1 | grep -v amdgpu.ids >> $O/ab_call18_wide_step0.txt
for lds in 1 0; do
  echo "RECOGYM_SWEEP_LDS=$lds" >> $O/ab_call18_wide_step0.txt
  RECOGYM_SWEEP_LDS=$lds timeout 200 python tools/wide_step0.py 1250000 2>&1 | grep -v amdgpu.ids >> $O/ab_call18_wide_step0.txt
done
run() { # name, env, args
  name=$1; envs=$2; shift; shift
  env $envs timeout 600 python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab18.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events=d['config']['events_per_step'], ctr=d['config']['ctr'], ms_per_step=round(d['ms_per_step'],2), value=d['value'], kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items()})))" | tee -a $O/ab_call18_c4.jsonl
}
rm -f $O/ab_call18_c4.jsonl
run c4shard_tpw "RECOGYM_SWEEP_LDS=1" --workload c4shard
