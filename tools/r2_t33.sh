timeout 200 python tools/step_probe.py c3 2>&1 | grep -v amdgpu.ids
