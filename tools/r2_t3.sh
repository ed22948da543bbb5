set -x
mkdir -p gpurun_out/r2_t3
python -m pytest tests -m gpu -q -x -k "sigma_omega_zero or sum_cache or certificate_is_sound or fixture or matches_oracle" 2>&1 | tail -15
timeout 200 python bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2_t3/c3.json 2> gpurun_out/r2_t3/c3.err; python - <<'PY'
import json
for f in ['gpurun_out/r2_t3/c3.json']:
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'])
PY
timeout 100 python bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2_t3/c2.json 2> gpurun_out/r2_t3/c2.err; python - <<'PY'
import json
for f in ['gpurun_out/r2_t3/c2.json']:
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'])
PY
