# Round 6, GPU call 59: the determinism probe on the paths the bench shapes do not take: lock-step (RECOGYM_RUN_AHEAD=0) on a C4 shard and on
# c3drift at 2 M users, a strongly scaled C3's shard (1.25 M users: hand-over at 16), C3 with the walk's pipeline off, C2.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
rm -f $O/determinism_call59.jsonl
RECOGYM_RUN_AHEAD=0 timeout 400 python tools/determinism_probe.py c4shard 3 0 0 400000 2>/dev/null | tail -1 >> $O/determinism_call59.jsonl
RECOGYM_RUN_AHEAD=0 timeout 400 python tools/determinism_probe.py c3drift 3 0 0 2000000 2>/dev/null | tail -1 >> $O/determinism_call59.jsonl
timeout 300 python tools/determinism_probe.py c3 4 0 0 1250000 2>/dev/null | tail -1 >> $O/determinism_call59.jsonl
RECOGYM_PIPE=0 timeout 300 python tools/determinism_probe.py c3 3 0 0 5000000 2>/dev/null | tail -1 >> $O/determinism_call59.jsonl
timeout 300 python tools/determinism_probe.py c3 3 1 0 5000000 2>/dev/null | tail -1 >> $O/determinism_call59.jsonl
cut -c1-330 $O/determinism_call59.jsonl
