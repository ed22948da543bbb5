for u in 1250000 2500000 5000000; do
timeout 300 python bench.py --workload c3 --users $u --steps 3 --warmup 1 --no-cpu-baseline --no-drift-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 users $u', round(d['value']/1e6,1), 'M ev/s', round(d['ms_per_step'],1), 'ms', {k:(v['ms']) for k,v in d['kernels'].items()})"
done
