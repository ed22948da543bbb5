mkdir -p gpurun_out/r2_t34
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "fp32_decided" 2>&1 | tail -2
for ab in 0 2097152 0 2097152; do
RECOGYM_ABLATE=$ab timeout 300 python bench.py --workload c3drift --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2_t34/d$ab.json 2> gpurun_out/r2_t34/d.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_t34/d$ab.json').read().strip().splitlines()[-1]); print('c3drift ablate $ab', round(d['value']/1e6,1), round(d['ms_per_step'],1), {k:(v['ms']) for k,v in d['kernels'].items()})
PY
done
