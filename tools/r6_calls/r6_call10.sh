# Round 6, GPU call 10: the wide sweep (k_draw_f16w, K = 64) with one / two user groups per wave, with and without its books
# (timing build, RECOGYM_ABLATE=256: results wrong by design) — is UG = 2 held back by the reloads behind its scratch stores?
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
rm -f $O/ab_call10_wide_step0.txt
for ug in 1 2; do for abl in 0 256 768 16640; do
  echo "UG=$ug ablate=$abl" >> $O/ab_call10_wide_step0.txt
  RECOGYM_HIP_LIB=$R/recogym_amd/csrc/librecogym_hip_f16wtiming.so RECOGYM_F16W_UG=$ug RECOGYM_ABLATE=$abl timeout 200 python tools/wide_step0.py 500000 >> $O/ab_call10_wide_step0.txt 2>&1
done; done
for ug in 1 2; do
  echo "default build UG=$ug" >> $O/ab_call10_wide_step0.txt
  RECOGYM_F16W_UG=$ug timeout 200 python tools/wide_step0.py 500000 >> $O/ab_call10_wide_step0.txt 2>&1
done
