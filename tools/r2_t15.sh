mkdir -p gpurun_out/r2_t15
timeout 1000 python -m pytest tests -m gpu -q 2>&1 | tail -6
for wl in c3 c2 c4shard; do
timeout 300 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2_t15/$wl.json 2> gpurun_out/r2_t15/$wl.err; tail -2 gpurun_out/r2_t15/$wl.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_t15/$wl.json').read().strip().splitlines()[-1]); print('$wl', round(d['value']/1e6,1), round(d['ms_per_step'],1), {k:(v['ms'],v['frac']) for k,v in d['kernels'].items()}, '| roofline', d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('traffic'))
if d.get('sigma_omega_gt0'): print('  drift', round(d['sigma_omega_gt0']['value']/1e6,1), {k:(v['ms'],v['frac']) for k,v in d['sigma_omega_gt0']['kernels'].items()})
PY
done
