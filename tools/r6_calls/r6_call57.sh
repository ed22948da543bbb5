# Round 6, GPU call 57: test_two_runs_of_a_bench_shape_give_the_same_log on a build with k_draw_tpw's bug put back (-DRG_TEST_TPW_LATE_BARRIER).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
RECOGYM_HIP_LIB=$R/recogym_amd/csrc/librecogym_hip_latebar.so timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "two_runs_of_a_bench_shape" 2>&1 | tail -12 | cut -c1-200 > $O/gpu_tests_call57_with_the_bug_put_back.txt
cat $O/gpu_tests_call57_with_the_bug_put_back.txt
