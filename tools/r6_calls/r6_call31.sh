# Round 6, GPU call 31: kernel-trace stats of C5 (which kernels the "logreg_acts" time is), fp16 and 8-bit screens.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in fp16 int8; do
  RECOGYM_LOGREG=$mode timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5stats_$mode -o run -- python $R/bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads > $O/c5stats_$mode.log 2> $O/c5stats_$mode.err
  f=$(find $O/c5stats_$mode -name '*kernel_stats.csv' | head -1); cp $f $O/c5_${mode}_kernel_stats_call31.csv; rm -rf $O/c5stats_$mode
  head -14 $O/c5_${mode}_kernel_stats_call31.csv | cut -c1-200
done
