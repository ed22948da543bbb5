# Round 5, GPU call 21: k_walk2's iterations by kind (tools/walk_kinds.py, -DRG_WALK_TIMING build): C3 with and without helpers.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
L=$R/recogym_amd/csrc/librecogym_hip_walktiming.so
rm -f $O/walk_kinds.jsonl
RECOGYM_HIP_LIB=$L timeout 200 python tools/walk_kinds.py c3 2>>$O/ab21.err | tail -1 >> $O/walk_kinds.jsonl
RECOGYM_HIP_LIB=$L RECOGYM_WALK_HELPERS=0 timeout 200 python tools/walk_kinds.py c3 2>>$O/ab21.err | tail -1 >> $O/walk_kinds.jsonl
RECOGYM_HIP_LIB=$L timeout 200 python tools/walk_kinds.py c2 2>>$O/ab21.err | tail -1 >> $O/walk_kinds.jsonl
tail -5 $O/ab21.err
