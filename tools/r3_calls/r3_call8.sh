# Round 3, GPU call 8: omega drift as its own kernel (a lane per user and Box-Muller pair) — parity + timings.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q --maxfail=12 -x > $O/gpu_tests8.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests8.log; tail -30 $O/gpu_tests8.log | cut -c1-300
B="--steps 1 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py $B $WL 2>$O/ab8_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events_per_s=round(d['value']/1e6,1), ms_per_step=round(d['ms_per_step'],2), kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" >> $O/ab8.jsonl
}
rm -f $O/ab8.jsonl
WL="--workload c3drift"
run c3drift A=1
WL="--workload c5"
run c5 A=1
WL="--workload c4shard"
run c4shard A=1
B="--steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
WL="--workload c3"
run c3 A=1
cat $O/ab8.jsonl
timeout 900 python tools/full_scale_check.py c4shard > $O/full_scale_parity_call8.txt 2> $O/full_scale_parity_call8.err; echo "full_scale rc=$?"; grep -h "verdict" $O/full_scale_parity_call8.txt | cut -c1-400
