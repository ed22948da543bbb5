# Round 6, GPU call 36: evidence on the tree after the LogReg work — the whole GPU suite, smoke, the default bench command.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | tail -8 > $O/gpu_tests_call36.txt
cat $O/gpu_tests_call36.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_call36.txt 2>&1
tail -2 $O/smoke_call36.txt
timeout 1200 python bench.py > $O/bench_default_call36.json 2> $O/bench_default_call36.err
tail -c 1500 $O/bench_default_call36.json
