mkdir -p gpurun_out/r2_t23
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "sigma_zero or cache or walk or fixture or case" 2>&1 | tail -3
for wl in c3 c2; do
timeout 300 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2_t23/$wl.json 2> gpurun_out/r2_t23/$wl.err; tail -2 gpurun_out/r2_t23/$wl.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_t23/$wl.json').read().strip().splitlines()[-1]); print('$wl', round(d['value']/1e6,1), round(d['ms_per_step'],1), {k:(v['ms'],v['frac']) for k,v in d['kernels'].items()})
PY
done
