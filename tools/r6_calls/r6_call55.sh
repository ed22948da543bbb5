# Round 6, GPU call 55: the determinism probe on C5's LogReg arm (the rebuilt act) and on C5 with the reference-fitted policies.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
rm -f $O/determinism_call55.jsonl
timeout 300 python tools/determinism_probe.py c5 5 0 1 2>/dev/null | tail -1 >> $O/determinism_call55.jsonl
timeout 300 python tools/determinism_probe.py c5trained 4 0 0 2>/dev/null | tail -1 >> $O/determinism_call55.jsonl
timeout 300 python tools/determinism_probe.py c5trained 4 0 1 2>/dev/null | tail -1 >> $O/determinism_call55.jsonl
cut -c1-420 $O/determinism_call55.jsonl
