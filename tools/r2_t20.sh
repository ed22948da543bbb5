mkdir -p gpurun_out/r2_t20
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "every_K_class" > gpurun_out/r2_t20/a.log 2>&1; tail -2 gpurun_out/r2_t20/a.log
RECOGYM_POISON=1 timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -k "not full_size and not million" > gpurun_out/r2_t20/b.log 2>&1; tail -12 gpurun_out/r2_t20/b.log
