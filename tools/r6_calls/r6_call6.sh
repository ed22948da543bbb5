# Round 6, GPU call 6: k_pick with the next group's loads in flight (parity, c3drift / C5 lines); where k_draw_tp's and k_pick's time
# goes — timing builds (-DRG_TP_ABL / -DRG_PICK_ABL bits, results wrong by design) on one unsliced step of 2 M users.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "lds_search or (certificate_is_sound_for_uniforms and f16_lds) or fused_and_sliced or run_ahead_rounds" 2>&1 | tail -8 > $O/gpu_tests_call6.txt
rm -f $O/ab_call6_tp_ablation.jsonl
for v in default tpabl1 tpabl2 tpabl3 tpabl4 tpabl8 tpabl16 tpabl32 tpabl64 tpabl125 pkabl1 pkabl2 pkabl4 pkocc3 pkocc2; do
  lib=$R/recogym_amd/csrc/librecogym_hip_$v.so
  [ $v = default ] && lib=$R/recogym_amd/csrc/librecogym_hip.so
  RECOGYM_HIP_LIB=$lib timeout 120 python tools/tp_probe.py 2000000 $v 2>>$O/ab6.err | tail -1 >> $O/ab_call6_tp_ablation.jsonl
done
run() { # name, env, args
  name=$1; envs=$2; shift; shift
  env $envs timeout 300 python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab6.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events=d['config']['events_per_step'], ctr=d['config']['ctr'], ms_per_step=round(d['ms_per_step'],2), value=d['value'], kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items()})))" | tee -a $O/ab_call6_tp.jsonl
}
rm -f $O/ab_call6_tp.jsonl
run c3drift_tp "RECOGYM_SWEEP_LDS=1" --workload c3drift
run c3drift_tp_occ3 "RECOGYM_HIP_LIB=$R/recogym_amd/csrc/librecogym_hip_pkocc3.so" --workload c3drift
run c5_tp "RECOGYM_SWEEP_LDS=1" --workload c5
