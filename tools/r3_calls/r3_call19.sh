# Round 3, GPU call 19: k_walk2 — lanes waiting for ctr per batch (RECOGYM_WALK_CLICK_BATCH), timing experiments that keep the
# trajectories (no log rows / no history insertion / no policy act; -DRG_WALK_TIMING build, results wrong by design).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py $B $WL 2>$O/ab19_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(json.dumps(dict(name='$name', ms_per_step=round(d['ms_per_step'],2), round1_ms=r.get('round1_ms'), later_rounds_ms=r.get('later_rounds_ms'), kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" >> $O/ab19.jsonl
}
rm -f $O/ab19.jsonl
B="--steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
WL="--workload c3"
for b in 4 0 1 2 8 16; do run click_batch_$b RECOGYM_WALK_CLICK_BATCH=$b; done
L=$R/recogym_amd/csrc/librecogym_hip_timing.so
run timing_build RECOGYM_HIP_LIB=$L
run no_log_rows RECOGYM_HIP_LIB=$L RECOGYM_ABLATE=$((1<<25))
run no_history_insert RECOGYM_HIP_LIB=$L RECOGYM_ABLATE=$((1<<27))
run no_policy_act RECOGYM_HIP_LIB=$L RECOGYM_ABLATE=$((1<<28))
run no_rows_insert_act RECOGYM_HIP_LIB=$L RECOGYM_ABLATE=$(((1<<25)+(1<<27)+(1<<28)))
WL="--workload c2"
for b in 4 0 2 8; do run c2_click_batch_$b RECOGYM_WALK_CLICK_BATCH=$b; done
cat $O/ab19.jsonl
