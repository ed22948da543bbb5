mkdir -p gpurun_out/r2_t9
for occ in 2 3 4; do
RECOGYM_WALK_OCC=$occ timeout 200 python bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline --no-drift-line > gpurun_out/r2_t9/c3_$occ.json 2> gpurun_out/r2_t9/c3_$occ.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_t9/c3_$occ.json').read().strip().splitlines()[-1]); print('occ $occ', round(d['ms_per_step'],1), {k:v['ms'] for k,v in d['kernels'].items()}, 'walk', d['roofline'].get('tail_ms'))
PY
done
