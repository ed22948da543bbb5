mkdir -p gpurun_out/r2_t11
timeout 900 python -m pytest tests -m gpu -q -x -k "K_class or matches_oracle or fixture or certificate_is_sound or wide_logit or every_draw_kernel" 2>&1 | tail -8
timeout 300 python bench.py --workload c4shard --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2_t11/c4.json 2> gpurun_out/r2_t11/c4.err; tail -3 gpurun_out/r2_t11/c4.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_t11/c4.json').read().strip().splitlines()[-1]); print('c4', d['value']/1e6, round(d['ms_per_step'],1), {k:(v['ms'],v['frac'],v.get('achieved')) for k,v in d['kernels'].items()}, d['roofline'].get('tail_ms'), d['roofline'].get('exact_fraction'))
PY
timeout 200 python bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline --no-drift-line > gpurun_out/r2_t11/c3.json 2> gpurun_out/r2_t11/c3.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_t11/c3.json').read().strip().splitlines()[-1]); print('c3', d['value']/1e6, round(d['ms_per_step'],1), {k:v['ms'] for k,v in d['kernels'].items()}, 'walk', d['roofline'].get('tail_ms'))
PY
