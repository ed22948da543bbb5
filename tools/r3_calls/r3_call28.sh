# Round 3, GPU call 28: final code (lock-step search from the chunk-major copy of Gamma) — the whole GPU suite, row-level parity
# at full size of the lock-step workloads it touches (C3 with drift, both C5 arms), the default bench line, C5.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q > $O/gpu_tests28.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests28.log; tail -3 $O/gpu_tests28.log | cut -c1-300
timeout 300 python tools/full_scale_check.py c3drift c5 > $O/full_scale_28.txt 2>&1; echo "full scale rc=$?"; grep verdict $O/full_scale_28.txt
timeout 300 python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()}, d['roofline']['frac'], d['roofline'].get('traffic_bytes_per_unit'), d.get('sigma_omega_gt0',{}).get('value'), {k:v['ms'] for k,v in d.get('sigma_omega_gt0',{}).get('kernels',{}).items()})" | cut -c1-900
timeout 120 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line > $O/bench_c5.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_c5.json').read().strip().splitlines()[-1]); print('c5', d['value'], d['ms_per_step'])"
