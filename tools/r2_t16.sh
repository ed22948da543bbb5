timeout 900 python -m pytest tests -m gpu -q -x -k "normal_time or fixture or dropin or generate_logs or per_user" 2>&1 | tail -12
