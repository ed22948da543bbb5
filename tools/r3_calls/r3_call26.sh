# Round 3, GPU call 26: sigma_omega > 0 — the search as its own kernel (k_draw_search over the whole step) instead of at the end of
# every user tile of the sweep; three or two waves per SIMD for it.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "lockstep or certificate_is_sound or every_K or drift or golden or reference or bandit_mf or logreg or shard" > $O/gpu_tests26.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests26.log; tail -4 $O/gpu_tests26.log | cut -c1-400
run() { # name, env...
  name=$1; shift
  env "$@" timeout 120 python bench.py $B $WL 2>$O/ab26_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events_per_s=round(d['value']/1e6,1), ms_per_step=round(d['ms_per_step'],2), kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" >> $O/ab26.jsonl
}
rm -f $O/ab26.jsonl
B="--steps 1 --warmup 1 --users 4000000 --no-cpu-baseline --no-drift-line --no-materialise"
WL="--workload c3drift"
run fused RECOGYM_SPLIT_SEARCH_OFF=1
run split_occ3 A=1
run split_occ2 RECOGYM_HIP_LIB=$R/recogym_amd/csrc/librecogym_hip_occ2.so
B="--steps 1 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
WL="--workload c5"
run c5_fused RECOGYM_SPLIT_SEARCH_OFF=1
run c5_split A=1
cat $O/ab26.jsonl
