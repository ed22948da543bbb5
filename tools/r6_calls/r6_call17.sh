# Round 6, GPU call 17: the shader clock under k_draw_tpw's load (s_memtime against the 100 MHz s_memrealtime over one work item),
# with and without its MFMAs.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
rm -f $O/ab_call17_tpw_clock.txt
for v in tpwclk tpwclk_nomfma; do
  echo "$v" >> $O/ab_call17_tpw_clock.txt
  RECOGYM_HIP_LIB=$R/recogym_amd/csrc/librecogym_hip_$v.so timeout 200 python tools/wide_step0.py 500000 2>&1 | grep -v amdgpu.ids >> $O/ab_call17_tpw_clock.txt
done
