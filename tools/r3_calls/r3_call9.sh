# Round 3, GPU call 9: the evidence the bench line cites — kernel-trace stats, PMC passes on --single-run, the default bench line.
R=$GRAFT_REPO_ROOT
cd $R
bash tools/r3_profiles.sh > gpurun_out/r3_prof.log 2>&1
tail -30 gpurun_out/r3_prof.log
cd $R
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r3/bench_default.json 2> gpurun_out/r3/bench_default.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/r3/bench_default.json
