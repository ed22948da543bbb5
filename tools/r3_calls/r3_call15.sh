# Round 3, GPU call 15: evidence with the final code — the whole GPU suite, row-level parity at full size of the lock-step
# workloads (C4 shard, both arms of C5, C3 with drift), the round's profiles (tools/r3_profiles.sh), the default bench line.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/gpu_tests15.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests15.log; tail -4 $O/gpu_tests15.log | cut -c1-300
timeout 900 python tools/full_scale_check.py c4shard c5 c3drift > $O/full_scale_15.txt 2>&1; echo "full scale rc=$?"; grep verdict $O/full_scale_15.txt
bash tools/r3_profiles.sh > $O/profiles15.log 2>&1; echo "profiles rc=$?"
cd $R
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline'], d.get('sigma_omega_gt0'), d['cpu_baseline'])" | cut -c1-2500
