timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3
bash tools/r2_profiles.sh > gpurun_out/r2_prof.log 2>&1; tail -2 gpurun_out/r2_prof.log
mkdir -p gpurun_out/r2_final
timeout 600 python bench.py > gpurun_out/r2_final/bench_default.json 2> gpurun_out/r2_final/bench_default.err; tail -c 300 gpurun_out/r2_final/bench_default.json
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
