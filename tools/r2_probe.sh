timeout 120 python tools/wide_step0.py 1250000 100000 2>&1 | grep -v amdgpu | tail -1
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "hip_matches_oracle or every_K or fixture or boundaries or every_draw_kernel" 2>&1 | tail -2
