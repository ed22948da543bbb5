# Round 6, GPU call 41: k_draw_tp's exp sums with plain v_add_f32 (alone this time: call 8 had them together with deferred exps).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
rm -f $O/ab_call41_tp_adds.jsonl
timeout 300 python tools/tp_probe.py 2000000 packed_adds 2>/dev/null | tail -1 >> $O/ab_call41_tp_adds.jsonl
RECOGYM_HIP_LIB=$R/recogym_amd/csrc/librecogym_hip_tpsc.so timeout 300 python tools/tp_probe.py 2000000 plain_adds 2>/dev/null | tail -1 >> $O/ab_call41_tp_adds.jsonl
timeout 300 python tools/tp_probe.py 2000000 packed_adds_b 2>/dev/null | tail -1 >> $O/ab_call41_tp_adds.jsonl
RECOGYM_HIP_LIB=$R/recogym_amd/csrc/librecogym_hip_tpsc.so timeout 300 python tools/tp_probe.py 2000000 plain_adds_b 2>/dev/null | tail -1 >> $O/ab_call41_tp_adds.jsonl
cat $O/ab_call41_tp_adds.jsonl
