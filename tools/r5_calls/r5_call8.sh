# Round 5, GPU call 8: the batched episode path for arbitrary Python agents (B users per rg_sim_step launch) — parity with the
# per-user path / the oracle / the host-path fixtures, and its events/s next to the per-user path's.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_env_dropin.py tests/test_host_logic.py -x -q -m gpu 2>&1 | tail -12 > $O/gpu_tests_call8.txt
timeout 600 python tools/per_user_path.py --users 4096 > $O/per_user_path.jsonl 2>$O/ab8.err
