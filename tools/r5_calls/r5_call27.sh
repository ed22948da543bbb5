# Round 5, GPU call 27: fp32 chunk recompute with its own budget (default build) against the float64 recompute
# (-DRG_WALK_PRECISE_CHUNK=1) on one box, C3, alternating.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
rm -f $O/ab_call27.jsonl
run() {  # name, workload, env...
  name=$1; wl=$2; shift; shift
  env "$@" timeout 150 python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab27.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events=d['config'].get('events_per_step'), ctr=d['config'].get('ctr'), ms_per_step=round(d['ms_per_step'],2), kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items()})))" >> $O/ab_call27.jsonl
}
L=$R/recogym_amd/csrc/librecogym_hip_precise.so
for i in 1 2 3; do
run fp32_chunk c3
run f64_chunk c3 RECOGYM_HIP_LIB=$L
done
