# Round 6, GPU call 8: k_draw_tp re-scheduled — an exp consumed a slot later, plain v_add_f32 (asm) instead of v_pk_add_f32, trees in the last slot.
# 2 / 3 / 4 blocks per CU (3: no spills, the default).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "lds_search or (certificate_is_sound_for_uniforms and f16_lds) or fused_and_sliced or run_ahead_rounds" 2>&1 | tail -8 > $O/gpu_tests_call8.txt
rm -f $O/ab_call8_tp_probe.jsonl
for v in default; do
  lib=$R/recogym_amd/csrc/librecogym_hip_$v.so
  [ $v = default ] && lib=$R/recogym_amd/csrc/librecogym_hip.so
  RECOGYM_HIP_LIB=$lib timeout 120 python tools/tp_probe.py 2000000 $v 2>>$O/ab7.err | tail -1 >> $O/ab_call8_tp_probe.jsonl
done
run() { # name, env, args
  name=$1; envs=$2; shift; shift
  env $envs timeout 300 python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab7.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events=d['config']['events_per_step'], ctr=d['config']['ctr'], ms_per_step=round(d['ms_per_step'],2), value=d['value'], kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items()})))" | tee -a $O/ab_call8_tp.jsonl
}
rm -f $O/ab_call8_tp.jsonl
run c3drift_tp "RECOGYM_SWEEP_LDS=1" --workload c3drift
run c5_tp "RECOGYM_SWEEP_LDS=1" --workload c5
