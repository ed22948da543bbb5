# Round 6, GPU call 49: final evidence — the whole GPU suite, smoke, the default bench command on the final tree.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | tail -8 > $O/gpu_tests_call49.txt
cat $O/gpu_tests_call49.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_call49.txt 2>&1
tail -2 $O/smoke_call49.txt
timeout 1200 python bench.py > $O/bench_default_call49.json 2> $O/bench_default_call49.err
tail -c 1500 $O/bench_default_call49.json
cd /tmp && export TMPDIR=/tmp
for w in c3 c3drift c5 c4shard; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -o run -- python $R/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads > $O/stats_$w.log 2> $O/stats_$w.err
  f=$(find $O/stats_$w -name '*kernel_stats.csv' | head -1); cp $f $O/${w}_kernel_stats_call49.csv; rm -rf $O/stats_$w
  grep '"metric"' $O/stats_$w.log > $O/${w}_bench_line_call49.json; rm -f $O/stats_$w.log
  head -6 $O/${w}_kernel_stats_call49.csv | cut -c1-150
done
