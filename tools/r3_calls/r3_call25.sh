# Round 3, GPU call 25: what the chunk-sum stores and the search cost the sigma_omega > 0 sweep (k_draw_bf16p; -DRG_SWEEP_TIMING
# build, results wrong by design): the upper bound of what a two-pass form without the 1.5 KB of chunk sums per draw could gain.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
run() { # name, env...
  name=$1; shift
  env "$@" timeout 120 python bench.py $B $WL 2>$O/ab25_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', ms_per_step=round(d['ms_per_step'],2), kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" >> $O/ab25.jsonl
}
rm -f $O/ab25.jsonl
B="--steps 1 --warmup 1 --users 4000000 --no-cpu-baseline --no-drift-line --no-materialise"
WL="--workload c3drift"
L=$R/recogym_amd/csrc/librecogym_hip_sweeptiming.so
run default_build A=1
run timing_build RECOGYM_HIP_LIB=$L
run no_chunk_sum_stores RECOGYM_HIP_LIB=$L RECOGYM_ABLATE=16
run no_search RECOGYM_HIP_LIB=$L RECOGYM_ABLATE=128
run no_stores_no_search RECOGYM_HIP_LIB=$L RECOGYM_ABLATE=144
run no_bookkeeping RECOGYM_HIP_LIB=$L RECOGYM_ABLATE=256
cat $O/ab25.jsonl
