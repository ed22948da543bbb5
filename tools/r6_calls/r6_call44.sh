# Round 6, GPU call 44: the highest user ids through the tiled sort (its table's empty key is the highest id), the RCCL test, the many-acts LogReg test.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_env_dropin.py -q -m gpu -k "highest_user_ids or rccl or more_acts_than or first_user or shards" 2>&1 | tail -5 > $O/gpu_tests_call44.txt
cat $O/gpu_tests_call44.txt
