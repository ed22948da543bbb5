# Round 6, GPU call 20: evidence on the final tree — the whole GPU suite, smoke, rocprofv3 kernel stats and PMC passes
# (tools/r6_profiles.sh), the default bench command.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests/ -q -m gpu 2>&1 | tail -8 > $O/gpu_tests_call20.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_call20.txt 2>&1
timeout 3000 bash tools/r6_profiles.sh > $O/r6_profiles.log 2>&1
