# Round 6, GPU call 51: call 50's full-size oracle replay found ONE wrong organic draw in a C4 shard (user 978356, event 232, second of two
# runs).  Which kernel: the same shard run 6 times per configuration, every ordered log against the first (tools/determinism_probe.py).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
rm -f $O/determinism_call51.jsonl
timeout 300 python tools/determinism_probe.py c4shard 6 1 2>/dev/null | tail -1 >> $O/determinism_call51.jsonl
RECOGYM_SWEEP_LDS=0 timeout 300 python tools/determinism_probe.py c4shard 6 1 2>/dev/null | tail -1 >> $O/determinism_call51.jsonl
RECOGYM_TAIL=128 timeout 300 python tools/determinism_probe.py c4shard 6 1 2>/dev/null | tail -1 >> $O/determinism_call51.jsonl
RECOGYM_RUN_AHEAD=0 timeout 400 python tools/determinism_probe.py c4shard 4 1 2>/dev/null | tail -1 >> $O/determinism_call51.jsonl
cut -c1-700 $O/determinism_call51.jsonl
