# Round 3, GPU call 2: k_walk2 (prefix sums + memo + LDS history) — parity suite, A/B timings against k_walk, full-size parity.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q --maxfail=12 > $O/gpu_tests2.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests2.log; tail -40 $O/gpu_tests2.log | cut -c1-300
B="--steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
run() { # name, env..., then bench args after --
  name=$1; shift
  env "$@" timeout 300 python bench.py $B $WL 2>$O/ab_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events_per_s=d['value'], ms_per_step=d['ms_per_step'], kernels={k:(v['ms']) for k,v in d['kernels'].items()}, walk=d['kernels'].get('walk'))))" >> $O/ab_walk.jsonl
}
rm -f $O/ab_walk.jsonl
WL="--workload c3"
run c3_old RECOGYM_WALK=1
run c3_w2_occ3 RECOGYM_WALK_OCC=3
run c3_w2_occ4 RECOGYM_WALK_OCC=4
run c3_w2_occ4_bias4 RECOGYM_WALK_OCC=4 RECOGYM_WALK_BIAS=4
run c3_w2_occ4_bias16 RECOGYM_WALK_OCC=4 RECOGYM_WALK_BIAS=16
WL="--workload c2"
run c2_old RECOGYM_WALK=1
run c2_w2_occ3 RECOGYM_WALK_OCC=3
run c2_w2_occ4 RECOGYM_WALK_OCC=4
cat $O/ab_walk.jsonl | cut -c1-400
timeout 600 python tools/full_scale_check.py c3 c2 > $O/full_scale_parity_walk2.txt 2> $O/full_scale_parity_walk2.err; echo "full_scale rc=$?"; grep verdict $O/full_scale_parity_walk2.txt
