# Round 5, GPU call 30: the default bench command once more with the PMC figures of call 29 in profiles/r5/pmc_traffic.json (the
# `issue_roofline` of call 29's line still priced round 4's instruction count), and the helpers test with the frozen table's float32
# propensities compared at float32 resolution (the five failures of call 29 were the TEST's tolerance: every index column matched).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "walk_helpers" 2>&1 | tail -3 > $O/gpu_tests_call30.txt
timeout 600 python bench.py > $O/c3_bench_line_call30.json 2> $O/c3_bench30.err; echo "bench rc=$?" >> $O/gpu_tests_call30.txt
