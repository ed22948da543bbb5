# Round 3, GPU call 10: LogReg screen in 20 one-block ranges; walk parameter sweep; the default bench line.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -x -k "logreg or verify_agents or frozen" > $O/gpu_tests10.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests10.log; tail -5 $O/gpu_tests10.log | cut -c1-300
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py $B $WL 2>$O/ab10_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events_per_s=round(d['value']/1e6,1), ms_per_step=round(d['ms_per_step'],2), kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" >> $O/ab10.jsonl
}
rm -f $O/ab10.jsonl
B="--steps 1 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
WL="--workload c5"
run c5 A=1
B="--steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
WL="--workload c3"
run c3_default A=1
run c3_bias4 RECOGYM_WALK_BIAS=4
run c3_bias16 RECOGYM_WALK_BIAS=16
run c3_refill4 RECOGYM_WALK_REFILL=4
run c3_refill16 RECOGYM_WALK_REFILL=16
run c3_mix6 RECOGYM_EXACT_MIX=6
run c3_mix4 RECOGYM_EXACT_MIX=4
cat $O/ab10.jsonl
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline'], d['materialise'], d['cpu_baseline'])" | cut -c1-1500
