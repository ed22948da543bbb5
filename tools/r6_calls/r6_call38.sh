# Round 6, GPU call 38: rg_sim_sort_log's scatter through LDS tiles (runs per user written contiguously) against the plain scatter:
# parity (fixtures, oracle cases, log invariants), then the sort of the whole C3 log both ways, c3drift's log too.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_env_dropin.py -q -m gpu -x -k "fixture or oracle or invariants or generate_logs or beyond_2_31" 2>&1 | tail -4 > $O/gpu_tests_call38.txt
cat $O/gpu_tests_call38.txt
timeout 600 python tools/sort_probe.py 10000000 > $O/sort_probe_call38_tiled.txt 2>/dev/null; cat $O/sort_probe_call38_tiled.txt
RECOGYM_SORT_PLAIN=1 timeout 600 python tools/sort_probe.py 10000000 > $O/sort_probe_call38_plain.txt 2>/dev/null; cat $O/sort_probe_call38_plain.txt
