mkdir -p gpurun_out/r2_t8
for ab in 0 65536 131072 262144 524288 1048576 1966080; do
RECOGYM_ABLATE=$ab timeout 200 python bench.py --workload c3 --steps 1 --warmup 0 --no-cpu-baseline --no-drift-line > gpurun_out/r2_t8/c3_$ab.json 2> gpurun_out/r2_t8/c3_$ab.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_t8/c3_$ab.json').read().strip().splitlines()[-1]); print('ablate $ab', round(d['ms_per_step'],1), {k:v['ms'] for k,v in d['kernels'].items()}, 'walk', d['roofline'].get('tail_ms'))
PY
done
