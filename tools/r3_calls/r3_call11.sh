# Round 3, GPU call 11: the whole GPU suite (with the wave-per-user and memo/anchored tests), k_walk2 with the main path's
# DevSim fields pinned in registers (two field sets) against the default build, shard sizes of C3, the default bench line.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/gpu_tests11.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests11.log; tail -5 $O/gpu_tests11.log | cut -c1-300
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py $B $WL 2>$O/ab11_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events_per_s=round(d['value']/1e6,1), ms_per_step=round(d['ms_per_step'],2), kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" >> $O/ab11.jsonl
}
rm -f $O/ab11.jsonl
B="--steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
L=$R/recogym_amd/csrc
for WLN in c3 c2; do
  WL="--workload $WLN"
  run ${WLN}_default A=1
  run ${WLN}_pin1 RECOGYM_HIP_LIB=$L/librecogym_hip_pin1.so
  run ${WLN}_pin2 RECOGYM_HIP_LIB=$L/librecogym_hip_pin2.so
  run ${WLN}_default_again A=1
done
for u in 1250000 2500000 5000000; do
  WL="--workload c3 --users $u"
  run c3_shard_$u A=1
  run c3_shard_${u}_pin2 RECOGYM_HIP_LIB=$L/librecogym_hip_pin2.so
done
cat $O/ab11.jsonl
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline'], d['materialise'], d['cpu_baseline'])" | cut -c1-1500
