# Round 6, GPU call 23: timing builds of the FINAL k_draw_tpw (tail-less chunk step, VGPR-form MFMAs), each with the shader clock
# and cycles per tile printed (bit 64): 1 no exps, 2 no MFMAs, 4 no barrier / DMA, 8 no books, 16 no mu seeds, 32 no ring re-reads.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
rm -f $O/ab_call23_tpw_ablation.txt
for v in tpwabl64 tpwabl65 tpwabl66 tpwabl68 tpwabl72 tpwabl96 tpwabl111; do
  echo "$v" >> $O/ab_call23_tpw_ablation.txt
  RECOGYM_HIP_LIB=$R/recogym_amd/csrc/librecogym_hip_$v.so timeout 200 python tools/wide_step0.py 500000 2>&1 | grep -v amdgpu.ids | tail -2 >> $O/ab_call23_tpw_ablation.txt
done
