mkdir -p gpurun_out/r2_t25
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x 2>&1 | tail -3
for sub in 0 1; do
for wl in c3 c2; do
RECOGYM_WALK_BIAS=8 RECOGYM_SUB=$sub timeout 300 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2_t25/${wl}_$sub.json 2> gpurun_out/r2_t25/$wl.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_t25/${wl}_$sub.json').read().strip().splitlines()[-1]); print('sub $sub $wl', round(d['value']/1e6,1), round(d['ms_per_step'],1), {k:(v['ms'],v['frac']) for k,v in d['kernels'].items()})
PY
done
done
