# Round 3, GPU call 23: evidence with the final code — the whole GPU suite, row-level parity at full size (C2, C4 shard, both
# C5 arms; C3 and C3 with drift: calls 18 / 21), the round's profiles (tools/r3_profiles.sh), the default bench line.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/gpu_tests23.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests23.log; tail -4 $O/gpu_tests23.log | cut -c1-300
timeout 600 python tools/full_scale_check.py c2 c4shard c5 > $O/full_scale_23.txt 2>&1; echo "full scale rc=$?"; grep verdict $O/full_scale_23.txt
bash tools/r3_profiles.sh > $O/profiles23.log 2>&1; echo "profiles rc=$?"
cd $R
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()}, d['roofline']['frac'], d.get('sigma_omega_gt0',{}).get('value'), d['cpu_baseline'].get('value'))" | cut -c1-1500
for WLN in c2 c5 c4shard; do timeout 300 python bench.py --workload $WLN --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line > $O/bench_$WLN.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_$WLN.json').read().strip().splitlines()[-1]); print('$WLN', d['value'], d['ms_per_step'])"; done
