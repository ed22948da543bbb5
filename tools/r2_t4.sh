mkdir -p gpurun_out/r2_t4
for ab in 0 1024 2048 3072; do
RECOGYM_ABLATE=$ab timeout 200 python bench.py --workload c3 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r2_t4/c3_$ab.json 2> gpurun_out/r2_t4/c3_$ab.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_t4/c3_$ab.json').read().strip().splitlines()[-1]); print('ablate $ab', d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'])
PY
done
