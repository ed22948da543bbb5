mkdir -p gpurun_out/r2_t19
for i in 1 2 3; do
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "not full_size and not million" > gpurun_out/r2_t19/run$i.log 2>&1; tail -3 gpurun_out/r2_t19/run$i.log
done
