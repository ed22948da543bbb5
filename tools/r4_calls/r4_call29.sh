# Round 4, GPU call 29 (the round's last seconds of GPU time): k_draw_bf16p with ONE set of A fragments loaded in place, compiled for
# three blocks per CU (-DRG_SWEEP_ONE_SET=1 -DRG_SWEEP_OCC=3, a build of this call): smoke against the oracle, then C3.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
X=$R/recogym_amd/csrc/librecogym_hip_oneset3.so
RECOGYM_HIP_LIB=$X timeout 28 python bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise 2>$O/ab29.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='c3_one_set_three_blocks', events=d['config']['events_per_step'], ctr=d['config']['ctr'], ms_per_step=round(d['ms_per_step'],2), kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" | tee $O/ab_call29_sweep_one_set_three_blocks.jsonl
RECOGYM_HIP_LIB=$X timeout 25 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-160 | tee -a $O/ab_call29_sweep_one_set_three_blocks.jsonl
