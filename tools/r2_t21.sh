mkdir -p gpurun_out/r2_t21
timeout 300 python tools/k65_probe.py 2 9 11 15 16 21 22 27 32 40 65 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_t21/a.log
