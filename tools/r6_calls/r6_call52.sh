# Round 6, GPU call 52: the determinism probe on the NARROW LDS sweep (k_draw_tp + k_pick): c3drift and C5's table arm, 5 runs each.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
rm -f $O/determinism_call52.jsonl
timeout 400 python tools/determinism_probe.py c3drift 5 0 2>/dev/null | tail -1 >> $O/determinism_call52.jsonl
timeout 400 python tools/determinism_probe.py c5 5 0 2>/dev/null | tail -1 >> $O/determinism_call52.jsonl
cut -c1-500 $O/determinism_call52.jsonl
