# Round 5, GPU call 28: the float64 batch's matrix / vector mix (exact_mix: groups of every 8 in the matrix form) for 0.845 M parked
# users (round 4 tuned it for 2.64 M), and the walk's occupancy knob once more with helpers.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
rm -f $O/ab_call28.jsonl
run() {  # name, workload, env...
  name=$1; wl=$2; shift; shift
  env "$@" timeout 150 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab28.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events=d['config'].get('events_per_step'), ms_per_step=round(d['ms_per_step'],2), kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items()})))" >> $O/ab_call28.jsonl
}
run default c3
for m in 2 3 4 6 7 8; do run exact_mix_$m c3 RECOGYM_EXACT_MIX=$m; done
run default2 c3
