# Round 6, GPU call 37: rg_sim_sort_log on the whole C3 log — which of its kernels the 20.7 ms are.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sortstats -o run -- python $R/tools/sort_probe.py 10000000 > $O/sort_probe_call37.txt 2> $O/sort_probe_call37.err
f=$(find $O/sortstats -name '*kernel_stats.csv' | head -1); cp $f $O/sort_kernel_stats_call37.csv; rm -rf $O/sortstats
cat $O/sort_probe_call37.txt; grep "scatter\|k_scan\|k_rows_per\|copyBuffer\|fill" $O/sort_kernel_stats_call37.csv | cut -c1-150
