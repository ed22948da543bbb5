# Round 3, GPU call 6: prefix form written by the sweep itself (no conversion pass), solo row-reservation fix.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q --maxfail=12 -x > $O/gpu_tests6.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests6.log; tail -30 $O/gpu_tests6.log | cut -c1-300
B="--steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py $B $WL 2>$O/ab6_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events_per_s=round(d['value']/1e6,1), ms_per_step=round(d['ms_per_step'],2), kernels={k:(v['ms']) for k,v in d['kernels'].items()}, r1=d['kernels']['walk']['round1_ms'], later=d['kernels']['walk']['later_rounds_ms'])))" >> $O/ab6.jsonl
}
rm -f $O/ab6.jsonl
WL="--workload c3"
run c3_fused A=1
run c3_unfused RECOGYM_SWEEP_PREFIX_OFF=1
WL="--workload c3 --users 1250000"
run c3s_fused A=1
WL="--workload c2"
run c2_fused A=1
run c2_unfused RECOGYM_SWEEP_PREFIX_OFF=1
cat $O/ab6.jsonl
timeout 900 python tools/full_scale_check.py c3 c2 > $O/full_scale_parity_call6.txt 2> $O/full_scale_parity_call6.err; echo "full_scale rc=$?"; grep verdict $O/full_scale_parity_call6.txt
