# Round 3, GPU call 14: the float64 MFMA does not co-execute with float64 VALU work (ubench, call 13) — which mix of the
# matrix and the vector form is fastest for the float64 sums then?  Certificate tests with the final code.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "certificate or sum_cache or memo or wave_per_user or every_K or sigma_omega_zero" > $O/gpu_tests14.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests14.log; tail -5 $O/gpu_tests14.log | cut -c1-400
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py $B $WL 2>$O/ab14_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(json.dumps(dict(name='$name', events_per_s=round(d['value']/1e6,1), ms_per_step=round(d['ms_per_step'],2), exact_fraction=r.get('exact_fraction'), kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" >> $O/ab14.jsonl
}
rm -f $O/ab14.jsonl
B="--steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
WL="--workload c3"
for m in 5 0 2 3 4 6 8; do run c3_mix$m RECOGYM_EXACT_MIX=$m; done
B="--steps 1 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
WL="--workload c3drift"; run c3drift_default A=1; run c3drift_valu RECOGYM_EXACT=valu
WL="--workload c4shard"; run c4shard_default A=1; run c4shard_valu RECOGYM_EXACT=valu
cat $O/ab14.jsonl
