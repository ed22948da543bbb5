mkdir -p gpurun_out/r2_t22
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "every_K_class" > gpurun_out/r2_t22/a.log 2>&1; grep -a "mismatches\|passed\|failed\|Error" gpurun_out/r2_t22/a.log | tail -5
timeout 300 python tools/k65_probe.py 65 100 128 2>&1 | grep -v amdgpu.ids
