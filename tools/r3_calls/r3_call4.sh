# Round 3, GPU call 4: k_walk_solo (last round: a wave per user, a lane per consecutive event) — parity + A/B.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q --maxfail=12 -x > $O/gpu_tests4.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests4.log; tail -30 $O/gpu_tests4.log | cut -c1-300
B="--steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py $B $WL 2>$O/ab4_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events_per_s=round(d['value']/1e6,1), ms_per_step=round(d['ms_per_step'],2), kernels={k:(v['ms']) for k,v in d['kernels'].items()}, r1=d['kernels']['walk']['round1_ms'], later=d['kernels']['walk']['later_rounds_ms'])))" >> $O/ab4.jsonl
}
rm -f $O/ab4.jsonl
WL="--workload c3"
run c3_nosolo RECOGYM_WALK_SOLO=0
run c3_solo_h16 RECOGYM_WALK_HANDOVER=16
run c3_solo_h32 RECOGYM_WALK_HANDOVER=32
run c3_solo_h48 RECOGYM_WALK_HANDOVER=48
run c3_solo_h24 RECOGYM_WALK_HANDOVER=24
WL="--workload c3 --users 1250000"
run c3s_nosolo RECOGYM_WALK_SOLO=0
run c3s_solo_h16 RECOGYM_WALK_HANDOVER=16
run c3s_solo_h32 RECOGYM_WALK_HANDOVER=32
run c3s_solo_h48 RECOGYM_WALK_HANDOVER=48
WL="--workload c2"
run c2_nosolo RECOGYM_WALK_SOLO=0
run c2_solo_h16 RECOGYM_WALK_HANDOVER=16
run c2_solo_h32 RECOGYM_WALK_HANDOVER=32
run c2_solo_h48 RECOGYM_WALK_HANDOVER=48
cat $O/ab4.jsonl
timeout 600 python tools/full_scale_check.py c3 c2 > $O/full_scale_parity_solo.txt 2> $O/full_scale_parity_solo.err; echo "full_scale rc=$?"; grep verdict $O/full_scale_parity_solo.txt
