# Round 6, GPU call 60: the whole GPU suite and smoke on the final tree (after the determinism tests were added).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | tail -8 > $O/gpu_tests_call60.txt
cat $O/gpu_tests_call60.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_call60.txt 2>&1
tail -1 $O/smoke_call60.txt | cut -c1-200
