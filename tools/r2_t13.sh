mkdir -p gpurun_out/r2_t13
for ug in 1 2; do
RECOGYM_F16W_UG=$ug timeout 300 python bench.py --workload c4shard --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2_t13/c4_$ug.json 2> gpurun_out/r2_t13/c4_$ug.err; tail -2 gpurun_out/r2_t13/c4_$ug.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_t13/c4_$ug.json').read().strip().splitlines()[-1]); print('c4 ug$ug', d['value']/1e6, round(d['ms_per_step'],1), {k:(v['ms'],v['frac'],v.get('achieved')) for k,v in d['kernels'].items()}, d['roofline'].get('tail_ms'), d['roofline'].get('exact_fraction'))
PY
done
RECOGYM_FULL_GRID=1 timeout 300 python bench.py --workload c4shard --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2_t13/c4_full.json 2> gpurun_out/r2_t13/c4_full.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_t13/c4_full.json').read().strip().splitlines()[-1]); print('c4 fullgrid', round(d['ms_per_step'],1), {k:v['ms'] for k,v in d['kernels'].items()})
PY
timeout 200 python bench.py --workload c3drift --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2_t13/c3d.json 2> gpurun_out/r2_t13/c3d.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_t13/c3d.json').read().strip().splitlines()[-1]); print('c3drift', d['value']/1e6, round(d['ms_per_step'],1), {k:(v['ms'],v['frac']) for k,v in d['kernels'].items()}, d['roofline'].get('tail_ms'))
PY
RECOGYM_FULL_GRID=1 timeout 200 python bench.py --workload c3drift --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2_t13/c3df.json 2> gpurun_out/r2_t13/c3df.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_t13/c3df.json').read().strip().splitlines()[-1]); print('c3drift fullgrid', round(d['ms_per_step'],1), {k:v['ms'] for k,v in d['kernels'].items()})
PY
