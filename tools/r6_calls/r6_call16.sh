# Round 6, GPU call 16: k_draw_tpw with nothing outside the MFMA stream (books, next seeds, next A fragments in its issue slots; barrier mid-tile).
# super-tiles — parity on every class, the C4 shard with and without it, step 0 alone.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "lds_search or every_K_class or (hip_matches_oracle and (14 or 3 or 4))" 2>&1 | tail -12 > $O/gpu_tests_call16.txt
rm -f $O/ab_call16_wide_step0.txt
for lds in 1 0; do
  echo "RECOGYM_SWEEP_LDS=$lds" >> $O/ab_call16_wide_step0.txt
  RECOGYM_SWEEP_LDS=$lds timeout 200 python tools/wide_step0.py 1250000 >> $O/ab_call16_wide_step0.txt 2>&1
done
run() { # name, env, args
  name=$1; envs=$2; shift; shift
  env $envs timeout 600 python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab12.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events=d['config']['events_per_step'], ctr=d['config']['ctr'], ms_per_step=round(d['ms_per_step'],2), value=d['value'], kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items()})))" | tee -a $O/ab_call16_c4.jsonl
}
rm -f $O/ab_call16_c4.jsonl
run c4shard_tpw "RECOGYM_SWEEP_LDS=1" --workload c4shard
run c4shard_f16w "RECOGYM_SWEEP_LDS=0" --workload c4shard
