RECOGYM_HIP_LIB=$PWD/recogym_amd/csrc/librecogym_hip_timing.so timeout 280 python tools/wide_probe.py 100000 600000 2>&1 | grep -v amdgpu
