# Round 6, GPU call 40: RCCL itself on this one-GPU box: one rank, then two ranks on the same device (tools/rccl_probe.py).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout 120 python tools/rccl_probe.py 1 > $O/rccl_probe_1rank.txt 2>&1; echo "rc $?" >> $O/rccl_probe_1rank.txt; tail -3 $O/rccl_probe_1rank.txt
NCCL_DEBUG=WARN timeout 120 python tools/rccl_probe.py 2 > $O/rccl_probe_2ranks_one_device.txt 2>&1; echo "rc $?" >> $O/rccl_probe_2ranks_one_device.txt; tail -12 $O/rccl_probe_2ranks_one_device.txt
