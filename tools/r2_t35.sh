mkdir -p gpurun_out/r2_t35
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "hip_matches_oracle or every_K or fixture" 2>&1 | tail -2
timeout 300 python bench.py --workload c4shard --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2_t35/c4.json 2> gpurun_out/r2_t35/c4.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_t35/c4.json').read().strip().splitlines()[-1]); print('c4', round(d['value']/1e6,1), round(d['ms_per_step'],1), {k:(v['ms'],v['frac']) for k,v in d['kernels'].items()})
PY
