mkdir -p gpurun_out/r2_t27
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_env_dropin.py -m gpu -q -x 2>&1 | tail -3
for wl in c3 c2; do
timeout 300 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2_t27/$wl.json 2> gpurun_out/r2_t27/$wl.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_t27/$wl.json').read().strip().splitlines()[-1]); print('$wl', round(d['value']/1e6,1), round(d['ms_per_step'],1), {k:(v['ms']) for k,v in d['kernels'].items()})
if d.get('sigma_omega_gt0'): print('  drift', round(d['sigma_omega_gt0']['value']/1e6,1), {k:(v['ms']) for k,v in d['sigma_omega_gt0']['kernels'].items()})
PY
done
