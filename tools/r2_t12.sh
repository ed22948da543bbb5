mkdir -p gpurun_out/r2_t12
timeout 900 python -m pytest tests -m gpu -q -x -k "million or K_class or matches_oracle or fixture or certificate_is_sound or wide_logit or ps_all or test_agent" 2>&1 | tail -8
timeout 300 python bench.py --workload c4shard --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2_t12/c4.json 2> gpurun_out/r2_t12/c4.err; tail -3 gpurun_out/r2_t12/c4.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_t12/c4.json').read().strip().splitlines()[-1]); print('c4', d['value']/1e6, round(d['ms_per_step'],1), {k:(v['ms'],v['frac'],v.get('achieved')) for k,v in d['kernels'].items()}, d['roofline'].get('tail_ms'), d['roofline'].get('exact_fraction'))
PY
