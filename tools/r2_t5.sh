mkdir -p gpurun_out/r2_t5
python -m pytest tests -m gpu -q -x 2>&1 | tail -15
for wl in c3 c2; do
timeout 200 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2_t5/$wl.json 2> gpurun_out/r2_t5/$wl.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_t5/$wl.json').read().strip().splitlines()[-1]); print('$wl', d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['exact_fraction'])
PY
done
