timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "hip_matches_oracle or every_K or fixture or boundaries or every_draw_kernel" 2>&1 | tail -2
echo "$(timeout 280 python tools/wide_probe.py 100000 400000 2>&1 | grep -v amdgpu | grep 'P=' | cut -c1-170)"
timeout 300 python bench.py --workload c4shard --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4', round(d['value']/1e6,1), round(d['ms_per_step'],1), {k:(v['ms'],v['frac']) for k,v in d['kernels'].items()})"
