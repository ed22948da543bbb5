# Round 6, GPU call 39: k_draw_tpw's exp sums with plain v_add_f32 instead of v_pk_add_f32 (MI355X_MICROARCH.md: a packed fp32 add beside
# MFMAs costs ~13 cycles beyond its slot at one wave per SIMD): cycles per tile on the device, then the C4 shard line of both builds.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
rm -f $O/ab_call39_tpw_adds.txt
for v in tpw64 tpwsc64; do
  echo "$v" >> $O/ab_call39_tpw_adds.txt
  RECOGYM_HIP_LIB=$R/recogym_amd/csrc/librecogym_hip_$v.so timeout 200 python tools/wide_step0.py 500000 2>&1 | grep -v amdgpu.ids | tail -2 >> $O/ab_call39_tpw_adds.txt
done
cat $O/ab_call39_tpw_adds.txt
run() { # name, env, args
  name=$1; envs=$2; shift; shift
  env $envs timeout 600 python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab39.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events=d['config']['events_per_step'], ms_per_step=round(d['ms_per_step'],2), value=d['value'], kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items()})))" | tee -a $O/ab_call39_c4.jsonl
}
rm -f $O/ab_call39_c4.jsonl $O/ab39.err
run c4_packed_adds "X=1" --workload c4shard
run c4_plain_adds "RECOGYM_HIP_LIB=$R/recogym_amd/csrc/librecogym_hip_tpwsc.so" --workload c4shard
