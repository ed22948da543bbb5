# Round 5, GPU call 7: reco-gym-v0 on the device (env_kind = 1) — fixtures of the unmodified reference, the oracle, the class surface;
# the v1 fixture and oracle cases again (k_advance changed).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_env_dropin.py tests/test_abi.py -x -q -m gpu -k "reference_fixture or env0 or hip_matches_oracle or abi or step_protocol or per_user_gym" 2>&1 | tail -12 > $O/gpu_tests_call7.txt
