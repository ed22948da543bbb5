mkdir -p gpurun_out/r2_t30
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x 2>&1 | tail -3
timeout 300 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2_t30/c3.json 2> gpurun_out/r2_t30/c3.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_t30/c3.json').read().strip().splitlines()[-1]); print('c3', round(d['value']/1e6,1), round(d['ms_per_step'],1), {k:(v['ms'],v['frac']) for k,v in d['kernels'].items()})
s=d['sigma_omega_gt0']; print('  drift', round(s['value']/1e6,1), round(s['ms_per_step'],1), {k:(v['ms'],v.get('frac')) for k,v in s['kernels'].items()})
PY
timeout 300 python bench.py --workload c4shard --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2_t30/c4.json 2> gpurun_out/r2_t30/c4.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_t30/c4.json').read().strip().splitlines()[-1]); print('c4', round(d['value']/1e6,1), round(d['ms_per_step'],1), {k:(v['ms'],v['frac']) for k,v in d['kernels'].items()})
PY
