mkdir -p gpurun_out/r2_t24
for b in 12 16 32 128; do
RECOGYM_WALK_BIAS=$b timeout 300 python bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2_t24/c3_$b.json 2> gpurun_out/r2_t24/c3.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_t24/c3_$b.json').read().strip().splitlines()[-1]); print('bias $b', round(d['value']/1e6,1), round(d['ms_per_step'],1), {k:(v['ms'],v['frac']) for k,v in d['kernels'].items()})
PY
done
