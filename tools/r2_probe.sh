timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "sigma_omega_zero or walk or sum_cache or fp32_decided" 2>&1 | tail -2
for i in 1 2; do
timeout 300 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --no-drift-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', round(d['value']/1e6,1), round(d['ms_per_step'],1), {k:(v['ms']) for k,v in d['kernels'].items()})"
done
