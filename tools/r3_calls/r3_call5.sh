# Round 3, GPU call 5: joint Ahat bound (tighter certificate budget), LogReg screen split by class ranges, handover 32.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q --maxfail=12 -x > $O/gpu_tests5.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests5.log; tail -30 $O/gpu_tests5.log | cut -c1-300
B="--steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py $B $WL 2>$O/ab5_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events_per_s=round(d['value']/1e6,1), ms_per_step=round(d['ms_per_step'],2), kernels={k:(v['ms']) for k,v in d['kernels'].items()}, exact=d['roofline'].get('exact_fraction'), roofline={k:d['roofline'].get(k) for k in ('kernel','ms','units','achieved','frac','us_per_launch','float64_refined_acts')})))" >> $O/ab5.jsonl
}
rm -f $O/ab5.jsonl
WL="--workload c3"
run c3 A=1
WL="--workload c3 --users 1250000"
run c3s A=1
WL="--workload c2"
run c2 A=1
B="--steps 1 --warmup 1 --no-cpu-baseline --no-materialise"
WL="--workload c5"
run c5 A=1
WL="--workload c3drift"
run c3drift A=1
WL="--workload c4shard"
run c4shard A=1
cat $O/ab5.jsonl
timeout 900 python tools/full_scale_check.py c3 c2 c4shard > $O/full_scale_parity_call5.txt 2> $O/full_scale_parity_call5.err; echo "full_scale rc=$?"; grep verdict $O/full_scale_parity_call5.txt
