# Round 3, GPU call 17: where k_walk2's time goes — timing experiments (-DRG_WALK_TIMING build, results wrong by design).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py $B $WL 2>$O/ab17_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(json.dumps(dict(name='$name', ms_per_step=round(d['ms_per_step'],2), round1_ms=r.get('round1_ms'), later_rounds_ms=r.get('later_rounds_ms'), exact_fraction=r.get('exact_fraction'), kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" >> $O/ab17.jsonl
}
rm -f $O/ab17.jsonl
B="--steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
WL="--workload c3"
L=$R/recogym_amd/csrc/librecogym_hip_timing.so
run timing_build_no_ablation RECOGYM_HIP_LIB=$L
run memo_always_hits_no_row_read RECOGYM_HIP_LIB=$L RECOGYM_ABLATE=$((1<<23))
run no_beta_row RECOGYM_HIP_LIB=$L RECOGYM_ABLATE=$((1<<24))
run no_log_rows RECOGYM_HIP_LIB=$L RECOGYM_ABLATE=$((1<<25))
run no_history_write_through RECOGYM_HIP_LIB=$L RECOGYM_ABLATE=$((1<<26))
run no_history_insert RECOGYM_HIP_LIB=$L RECOGYM_ABLATE=$((1<<27))
run memo_and_beta RECOGYM_HIP_LIB=$L RECOGYM_ABLATE=$(((1<<23)+(1<<24)))
run all_five RECOGYM_HIP_LIB=$L RECOGYM_ABLATE=$(((1<<23)+(1<<24)+(1<<25)+(1<<26)+(1<<27)))
cat $O/ab17.jsonl
