# Round 5, GPU call 34: the whole GPU suite, smoke and the default bench command on the final tree (after call 33's readfirstlane_u64).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests_call34_full.txt 2>&1; echo "pytest rc=$?" >> $O/gpu_tests_call34_full.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke_call34.txt 2>&1
timeout 600 python bench.py > $O/c3_bench_line_call34.json 2> $O/c3_bench34.err; echo "bench rc=$?" >> $O/smoke_call34.txt
