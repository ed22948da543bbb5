# Round 5, GPU call 5: what in k_sweep_xh's bookkeeping costs a quarter of the kernel — timing builds (results wrong by design).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
rm -f $O/ab_call5_xh_books.jsonl
for v in 0 64 128 256 512 1024 1920 0; do
  lib=$R/recogym_amd/csrc/librecogym_hip_xhabl$v.so
  [ $v = 0 ] && lib=$R/recogym_amd/csrc/librecogym_hip.so
  RECOGYM_HIP_LIB=$lib timeout 90 python tools/xh_probe.py 2000000 abl$v 2>>$O/ab5.err | tail -1 >> $O/ab_call5_xh_books.jsonl
done
