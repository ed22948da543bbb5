# Round 4, GPU call 25: the tail threshold and the repack cadence were tuned for lock-step steps; with rounds (a third as many
# launches, each dearer) — C3 with drift at 4 M users and C5 with fitted policies: RECOGYM_TAIL, RECOGYM_REPACK.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py $B $WL 2>$O/ab25_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', workload=d['config']['workload'].split(':')[0], events_per_s=round(d['value']/1e6,1), ms_per_step=round(d['ms_per_step'],2), kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" >> $O/ab_call25_tail_repack_in_rounds.jsonl
}
rm -f $O/ab_call25_tail_repack_in_rounds.jsonl
B="--steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
WL="--workload c3drift --users 4000000"
run drift_tail4096 A=1
run drift_tail8192 RECOGYM_TAIL=8192
run drift_tail2048 RECOGYM_TAIL=2048
run drift_tail1024 RECOGYM_TAIL=1024
run drift_tail512 RECOGYM_TAIL=512
run drift_repack8 RECOGYM_REPACK=8
run drift_repack32 RECOGYM_REPACK=32
run drift_repack4 RECOGYM_REPACK=4
WL="--workload c5trained"
run c5t_tail4096 A=1
run c5t_tail1024 RECOGYM_TAIL=1024
run c5t_tail16384 RECOGYM_TAIL=16384
cat $O/ab_call25_tail_repack_in_rounds.jsonl
