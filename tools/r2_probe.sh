timeout 280 python tools/wide_probe.py 2>&1 | grep -v amdgpu
