mkdir -p gpurun_out/r2_t26
for ab in 65536 131072 262144 1048576 1310720; do
RECOGYM_ABLATE=$ab timeout 300 python bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line > gpurun_out/r2_t26/c3_$ab.json 2> gpurun_out/r2_t26/c3.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_t26/c3_$ab.json').read().strip().splitlines()[-1]); print('ablate $ab', round(d['value']/1e6,1), round(d['ms_per_step'],1), {k:(v['ms']) for k,v in d['kernels'].items()})
PY
done
