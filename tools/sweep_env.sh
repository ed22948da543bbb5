#!/bin/bash
# usage: tools/sweep_env.sh VAR v1 v2 ...   -> one bench.py line (value + kernel ms) per setting
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v timeout 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 > /tmp/sweep_line.json
  python - "$VAR" "$v" <<'PY'
import json, sys
d = json.load(open('/tmp/sweep_line.json'))
print(sys.argv[1], sys.argv[2], round(d['value'] / 1e6, 1), d['roofline']['kernel_ms'])
PY
done
