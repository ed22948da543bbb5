# Round 4, GPU call 12: a phase offset between the two blocks a CU holds in the lock-step sweep (RECOGYM_SWEEP_STAGGER = units of
# 3.4 us), so that one block's search (a chain of loads) runs beside the other's sweep: C3 with drift.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py $B $WL 2>$O/ab12_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events_per_s=round(d['value']/1e6,1), ms_per_step=round(d['ms_per_step'],2), kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" >> $O/ab_call12_sweep_stagger.jsonl
}
rm -f $O/ab_call12_sweep_stagger.jsonl
B="--steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --users 4000000"
WL="--workload c3drift"
run off A=1
run s8 RECOGYM_SWEEP_STAGGER=8
run s16 RECOGYM_SWEEP_STAGGER=16
run s24 RECOGYM_SWEEP_STAGGER=24
run s32 RECOGYM_SWEEP_STAGGER=32
cat $O/ab_call12_sweep_stagger.jsonl
