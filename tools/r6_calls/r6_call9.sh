# Round 6, GPU call 9: the whole GPU suite, smoke, and the default bench command on the tree with k_draw_tp + k_pick.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8 > $O/gpu_tests_call9.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_call9.txt 2>&1
timeout 900 python bench.py > $O/bench_default_call9.json 2> $O/bench_default_call9.err
tail -c 1500 $O/bench_default_call9.json
