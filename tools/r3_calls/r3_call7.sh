# Round 3, GPU call 7: float64-ANCHORED certificate (prefix sums of the parked users' float64 sums), cheaper sweep bookkeeping.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q --maxfail=12 -x > $O/gpu_tests7.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests7.log; tail -30 $O/gpu_tests7.log | cut -c1-300
B="--steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py $B $WL 2>$O/ab7_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events_per_s=round(d['value']/1e6,1), ms_per_step=round(d['ms_per_step'],2), kernels={k:(v['ms']) for k,v in d['kernels'].items()}, r1=d['kernels']['walk']['round1_ms'], later=d['kernels']['walk']['later_rounds_ms'], exact=d['roofline'].get('exact_fraction'))))" >> $O/ab7.jsonl
}
rm -f $O/ab7.jsonl
WL="--workload c3"
run c3 A=1
WL="--workload c3 --users 1250000"
run c3s A=1
WL="--workload c2"
run c2 A=1
cat $O/ab7.jsonl
timeout 900 python tools/full_scale_check.py c3 c2 > $O/full_scale_parity_call7.txt 2> $O/full_scale_parity_call7.err; echo "full_scale rc=$?"; grep -h "verdict\|anchored" $O/full_scale_parity_call7.txt | cut -c1-400
