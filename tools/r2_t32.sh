mkdir -p gpurun_out/r2_t32
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_env_dropin.py -m gpu -q -x 2>&1 | tail -3
run() { # tag env...
tag=$1; shift
env "$@" timeout 300 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --no-drift-line > gpurun_out/r2_t32/c3_$tag.json 2> gpurun_out/r2_t32/c3.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_t32/c3_$tag.json').read().strip().splitlines()[-1]); w=d['kernels']['walk']; print('$tag', round(d['value']/1e6,1), round(d['ms_per_step'],1), w['ms'], w['round1_ms'], w['round2_ms'], d['kernels']['draw_exact_f64']['ms'])
PY
}
run h16 X=1
run h0 RECOGYM_WALK_HANDOVER=0
run h32 RECOGYM_WALK_HANDOVER=32
run h8 RECOGYM_WALK_HANDOVER=8
timeout 300 python bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2_t32/c2.json 2> gpurun_out/r2_t32/c2.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_t32/c2.json').read().strip().splitlines()[-1]); print('c2', round(d['value']/1e6,1), round(d['ms_per_step'],1), {k:(v['ms']) for k,v in d['kernels'].items()})
PY
