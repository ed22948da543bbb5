for wl in c2 c3; do
timeout 300 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-drift-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', round(d['value']/1e6,1), round(d['ms_per_step'],2))"
done
timeout 300 python bench.py --workload c3 --steps 3 --warmup 0 --no-cpu-baseline --no-drift-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 warmup 0', round(d['value']/1e6,1), round(d['ms_per_step'],2))"
