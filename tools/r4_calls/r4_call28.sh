# Round 4, GPU call 28: the sampled-oracle tests (bench workloads at size, the oracle replaying sampled users) on the final default
# build — the group call 27 left out that runs through the sweep.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 170 python -m pytest tests -m gpu -q -x -k "sampled_oracle" > $O/gpu_tests28.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests28.log; tail -4 $O/gpu_tests28.log | cut -c1-400
