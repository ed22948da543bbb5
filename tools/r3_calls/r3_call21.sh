# Round 3, GPU call 21: k_walk2's longer view histories — a view of a product in the line stays in LDS, a product behind the line
# touches only entries >= 16 of the row; click batch 8.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q -x -k "sum_cache or memo or wave_per_user or sigma_omega_zero or walk_certificate or ouc or history or hist or shard or golden or reference_log" > $O/gpu_tests21.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests21.log; tail -4 $O/gpu_tests21.log | cut -c1-400
run() { # name, env...
  name=$1; shift
  env "$@" timeout 60 python bench.py $B $WL 2>$O/ab21_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(json.dumps(dict(name='$name', events_per_s=round(d['value']/1e6,1), ms_per_step=round(d['ms_per_step'],2), round1_ms=r.get('round1_ms'), later_rounds_ms=r.get('later_rounds_ms'), kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" >> $O/ab21.jsonl
}
rm -f $O/ab21.jsonl
B="--steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
for WLN in c3 c2; do WL="--workload $WLN"; run ${WLN} A=1; done
WL="--workload c3 --users 1250000"; run c3_shard_1250000 A=1
cat $O/ab21.jsonl
timeout 300 python tools/full_scale_check.py c3 > $O/full_scale_21.txt 2>&1; echo "full scale rc=$?"; grep verdict $O/full_scale_21.txt
