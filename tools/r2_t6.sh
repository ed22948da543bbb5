mkdir -p gpurun_out/r2_t6
python -m pytest tests -m gpu -q -x -k "sigma_omega_zero or sum_cache or user_major or fixture or matches_oracle or dropin or generate_logs" 2>&1 | tail -15
for wl in c3 c2; do
timeout 200 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line > gpurun_out/r2_t6/$wl.json 2> gpurun_out/r2_t6/$wl.err
tail -3 gpurun_out/r2_t6/$wl.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_t6/$wl.json').read().strip().splitlines()[-1]); print('$wl', d['value']/1e6, d['ms_per_step'], {k:(v['ms'],v['frac']) for k,v in d['kernels'].items()}, d['roofline'].get('tail_ms'))
PY
done
