timeout 900 python -m pytest tests -m gpu -q -x -k "logreg or frozen or verify_agents or feed_to_sklearn" 2>&1 | tail -6
timeout 300 python tools/logreg_probe.py 1000 200000 2>&1 | grep -v amdgpu.ids
timeout 600 python tools/logreg_probe.py 10000 200000 2>&1 | grep -v amdgpu.ids
