# Round 5, GPU call 15: LogregMulticlassIps(select_randomly = True) on the device (k_logreg_sample) — the reference's fixture, the oracle
# at P = 10 / 200 / 1024, every other frozen-LogReg case again.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_env_dropin.py tests/test_abi.py -x -q -m gpu -k "logreg or abi or reference_fixture" 2>&1 | tail -12 > $O/gpu_tests_call15.txt
