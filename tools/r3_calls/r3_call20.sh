# Round 3, GPU call 20: k_walk2's view-history insertion — the line in LDS (<= 15 products) against the longer histories (row in
# memory); -DRG_WALK_TIMING build, results wrong by design.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
run() { # name, env...
  name=$1; shift
  env "$@" timeout 60 python bench.py $B $WL 2>$O/ab20_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(json.dumps(dict(name='$name', ms_per_step=round(d['ms_per_step'],2), round1_ms=r.get('round1_ms'), later_rounds_ms=r.get('later_rounds_ms'), kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" >> $O/ab20.jsonl
}
rm -f $O/ab20.jsonl
B="--steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
WL="--workload c3"
L=$R/recogym_amd/csrc/librecogym_hip_timing.so
run default_build A=1
run timing_build RECOGYM_HIP_LIB=$L
run no_insert_into_line RECOGYM_HIP_LIB=$L RECOGYM_ABLATE=$((1<<29))
run no_insert_into_longer_history RECOGYM_HIP_LIB=$L RECOGYM_ABLATE=$((1<<26))
cat $O/ab20.jsonl
