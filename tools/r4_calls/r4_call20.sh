# Round 4, GPU call 20: k_advance_run specialised by policy class (the lookup / bounded-draw policies without the view-history
# code: 162 registers instead of 246, three waves per SIMD; -DRG_ADV_LIGHT_WAVES=4: four, 36 spilled) — parity tests that reach the
# rounds, then C5, C5 with fitted policies and the C4 shard.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "run_ahead or lock_step_to_the_end or repacked or reproduces_reference_fixture or matches_oracle or repack_and_tail" > $O/gpu_tests20.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests20.log; tail -5 $O/gpu_tests20.log | cut -c1-600
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py $B $WL 2>$O/ab20_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', workload=d['config']['workload'].split(':')[0], events_per_s=round(d['value']/1e6,1), ms_per_step=round(d['ms_per_step'],2), kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" >> $O/ab_call20_advance_run_by_policy_class.jsonl
}
rm -f $O/ab_call20_advance_run_by_policy_class.jsonl
B="--steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
for wl in c5 c5trained c4shard; do
  WL="--workload $wl"
  run ${wl}_light3 A=1
  run ${wl}_light4 RECOGYM_HIP_LIB=$R/recogym_amd/csrc/librecogym_hip_light4.so
done
cat $O/ab_call20_advance_run_by_policy_class.jsonl
