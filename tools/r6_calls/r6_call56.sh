# Round 6, GPU call 56: two runs of chip-filling shapes give the same log — the C4 shard's shape and the other k-step classes of k_draw_tpw
# (K = 30 / 45 / 64), c3drift, C3.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "two_runs_of_a_bench_shape" 2>&1 | tail -5 > $O/gpu_tests_call56.txt
cat $O/gpu_tests_call56.txt
