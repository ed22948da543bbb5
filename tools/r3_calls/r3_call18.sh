# Round 3, GPU call 18: no click below u = 0.97 (the range of ff) — the bandit events that cannot click read neither beta[a] nor
# omega (walk: the others wait for a batch); the view history line is written back when the lane lets go of the user.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q -x > $O/gpu_tests18.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests18.log; tail -4 $O/gpu_tests18.log | cut -c1-400
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py $B $WL 2>$O/ab18_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(json.dumps(dict(name='$name', events_per_s=round(d['value']/1e6,1), ms_per_step=round(d['ms_per_step'],2), round1_ms=r.get('round1_ms'), later_rounds_ms=r.get('later_rounds_ms'), kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" >> $O/ab18.jsonl
}
rm -f $O/ab18.jsonl
B="--steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
for WLN in c3 c2; do WL="--workload $WLN"; run ${WLN} A=1; done
WL="--workload c3 --users 1250000"; run c3_shard_1250000 A=1
B="--steps 1 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
for WLN in c3drift c5 c4shard; do WL="--workload $WLN"; run ${WLN} A=1; done
cat $O/ab18.jsonl
timeout 900 python tools/full_scale_check.py c3 c2 c3drift > $O/full_scale_18.txt 2>&1; echo "full scale rc=$?"; grep verdict $O/full_scale_18.txt
