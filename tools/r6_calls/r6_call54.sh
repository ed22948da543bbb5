# Round 6, GPU call 54: final evidence after the k_draw_tpw fix — the oracle's full-size replay of a C4 shard (both runs), the whole GPU
# suite, smoke, the default bench command, kernel-trace stats of the C4 shard.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
rm -f $O/oracle_spot_check_call54.jsonl
timeout 900 python tests/oracle_spot_check.py c4shard --out $O/oracle_spot_check_call54.jsonl > $O/oracle_spot_check_call54.txt 2>&1
tail -3 $O/oracle_spot_check_call54.txt | cut -c1-300
timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | tail -8 > $O/gpu_tests_call54.txt
cat $O/gpu_tests_call54.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_call54.txt 2>&1
tail -1 $O/smoke_call54.txt | cut -c1-200
timeout 1200 python bench.py > $O/bench_default_call54.json 2> $O/bench_default_call54.err
tail -c 600 $O/bench_default_call54.json
cd /tmp && export TMPDIR=/tmp
for w in c4shard; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -o run -- python $R/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads > $O/stats_$w.log 2> $O/stats_$w.err
  f=$(find $O/stats_$w -name '*kernel_stats.csv' | head -1); cp $f $O/${w}_kernel_stats_call54.csv; rm -rf $O/stats_$w
  grep '"metric"' $O/stats_$w.log > $O/${w}_bench_line_call54.json; rm -f $O/stats_$w.log
  head -6 $O/${w}_kernel_stats_call54.csv | cut -c1-150
done
