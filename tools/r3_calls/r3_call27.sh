# Round 3, GPU call 27: the lock-step search's chunk recompute from the chunk-major copy of Gamma (eight lanes per user) instead
# of the row-major gather; fused against split search with it.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "lockstep or certificate_is_sound or every_K or drift or golden or reference or bandit_mf or logreg or shard or tile_boundar or K_class" > $O/gpu_tests27.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests27.log; tail -4 $O/gpu_tests27.log | cut -c1-400
run() { # name, env...
  name=$1; shift
  env "$@" timeout 120 python bench.py $B $WL 2>$O/ab27_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events_per_s=round(d['value']/1e6,1), ms_per_step=round(d['ms_per_step'],2), kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" >> $O/ab27.jsonl
}
rm -f $O/ab27.jsonl
B="--steps 1 --warmup 1 --users 4000000 --no-cpu-baseline --no-drift-line --no-materialise"
WL="--workload c3drift"
run fused RECOGYM_SPLIT_SEARCH_OFF=1
run split A=1
cat $O/ab27.jsonl
