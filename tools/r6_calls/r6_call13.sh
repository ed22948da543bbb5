# Round 6, GPU call 13: k_draw_tpw after the list-capacity fix — parity on every class; where its step-0 time goes (timing builds,
# -DRG_TPW_ABL bits, results wrong by design).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_hip_parity.py -q -m gpu -k "lds_search or every_K_class or (hip_matches_oracle and (14 or 3 or 4))" 2>&1 | tail -12 > $O/gpu_tests_call13.txt
rm -f $O/ab_call13_tpw_ablation.txt
for v in default tpwabl1 tpwabl2 tpwabl4 tpwabl8 tpwabl16 tpwabl32 tpwabl63; do
  lib=$R/recogym_amd/csrc/librecogym_hip_$v.so
  [ $v = default ] && lib=$R/recogym_amd/csrc/librecogym_hip.so
  echo "$v" >> $O/ab_call13_tpw_ablation.txt
  RECOGYM_HIP_LIB=$lib timeout 200 python tools/wide_step0.py 500000 2>&1 | grep -v amdgpu.ids >> $O/ab_call13_tpw_ablation.txt
done
