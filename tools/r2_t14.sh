mkdir -p gpurun_out/r2_t14
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
for occ in 3 4; do
RECOGYM_WALK_OCC=$occ timeout 200 python bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline --no-drift-line > gpurun_out/r2_t14/c3_$occ.json 2> gpurun_out/r2_t14/c3_$occ.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_t14/c3_$occ.json').read().strip().splitlines()[-1]); print('c3 occ$occ', d['value']/1e6, round(d['ms_per_step'],1), {k:v['ms'] for k,v in d['kernels'].items()}, 'walk', d['roofline'].get('tail_ms'))
PY
done
timeout 100 python bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline --no-drift-line > gpurun_out/r2_t14/c2.json 2> gpurun_out/r2_t14/c2.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_t14/c2.json').read().strip().splitlines()[-1]); print('c2', d['value']/1e6, round(d['ms_per_step'],1), {k:v['ms'] for k,v in d['kernels'].items()}, 'walk', d['roofline'].get('tail_ms'))
PY
