# Round 4, GPU call 26: k_draw_bf16p with two product tiles in LDS instead of three (one in flight) and, on top, compiled for three
# blocks per CU (-DRG_SWEEP_NB=2 [-DRG_SWEEP_OCC=3]: 168 registers with ~8 scratch accesses per loop iteration, 48 KB of LDS) — builds
# of this call; the sweep's parity tests on the three-block build, then C3 and C3 with drift (4 M users) on all three.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
S3=$R/recogym_amd/csrc/librecogym_hip_sweep3.so
R2=$R/recogym_amd/csrc/librecogym_hip_ring2.so
RECOGYM_HIP_LIB=$S3 timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "every_draw_kernel or every_K_class or tile_boundaries or wide_logit_range or certificate_is_sound_for_uniforms_next_to_cdf or sigma_omega_zero_sum_cache or hip_matches_oracle or hip_reproduces" > $O/gpu_tests26.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests26.log; tail -5 $O/gpu_tests26.log | cut -c1-600
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py $B $WL 2>$O/ab26_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', workload=d['config']['workload'].split(':')[0], events_per_s=round(d['value']/1e6,1), ms_per_step=round(d['ms_per_step'],2), kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" >> $O/ab_call26_sweep_ring2_three_blocks.jsonl
}
rm -f $O/ab_call26_sweep_ring2_three_blocks.jsonl
B="--steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
WL="--workload c3"
run c3_default A=1
run c3_ring2 RECOGYM_HIP_LIB=$R2
run c3_ring2_three_blocks RECOGYM_HIP_LIB=$S3
WL="--workload c3drift --users 4000000"
run drift_default A=1
run drift_ring2 RECOGYM_HIP_LIB=$R2
run drift_ring2_three_blocks RECOGYM_HIP_LIB=$S3
cat $O/ab_call26_sweep_ring2_three_blocks.jsonl
