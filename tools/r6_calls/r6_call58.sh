# Round 6, GPU call 58: the determinism test incl. C5's LogReg arm at 262 144 users.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "two_runs_of_a_bench_shape" 2>&1 | tail -5 > $O/gpu_tests_call58.txt
cat $O/gpu_tests_call58.txt
