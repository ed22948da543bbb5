# Round 5, GPU call 25: helpers leave histories beyond the line to their owners (the row loop ran in half the bandit iterations);
# parity, C3 / C2, and the kernel trace of one C3 bench run (which launch takes what now).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "walk or memo or sum_cache or hip_matches_oracle or organic_only or phantom or ouc or history" 2>&1 | tail -4 > $O/gpu_tests_call25.txt
rm -f $O/ab_call25.jsonl
run() {  # name, workload, env...
  name=$1; wl=$2; shift; shift
  env "$@" timeout 150 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab25.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events=d['config'].get('events_per_step'), ctr=d['config'].get('ctr'), ms_per_step=round(d['ms_per_step'],2), kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items()})))" >> $O/ab_call25.jsonl
}
run c3 c3
run c3 c3
run c2 c2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt25
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt25 -o kt -- python $R/bench.py --workload c3 --steps 4 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads > /dev/null 2>>$O/ab25.err
f=$(find /tmp/kt25 -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -14 $f > $O/kernel_stats_call25.csv
