# Round 4, GPU call 7: k_exact_sums_h<5> compiled for four waves per SIMD (127 registers, -DRG_EXACT_H_OCC=4) against the default
# build (102 + 32 registers, three waves), and the matrix / vector mix of the float64 batch on both.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
X=$R/recogym_amd/csrc/librecogym_hip_xocc4.so
timeout 900 python tools/pipe_probe.py --reps 3 \
  occ3mix5:A=1 occ3mix4:RECOGYM_EXACT_MIX=4 occ3mix6:RECOGYM_EXACT_MIX=6 \
  occ4mix5:RECOGYM_HIP_LIB=$X occ4mix4:RECOGYM_HIP_LIB=$X,RECOGYM_EXACT_MIX=4 occ4mix6:RECOGYM_HIP_LIB=$X,RECOGYM_EXACT_MIX=6 occ4mix3:RECOGYM_HIP_LIB=$X,RECOGYM_EXACT_MIX=3 \
  > $O/ab_call7_exact_occupancy.jsonl 2> $O/ab_call7.err; echo "probe rc=$?"; tail -3 $O/ab_call7.err
python - <<'PY'
import json
for l in open('gpurun_out/r4/ab_call7_exact_occupancy.jsonl'):
    d = json.loads(l)
    print(d['config'], d['best_ms'], d['run_ms'], {k: d['profile'][k] for k in ('draw_exact_ms','walk1_ms','walk2_ms')})
PY
