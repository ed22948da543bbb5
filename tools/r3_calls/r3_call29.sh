# Round 3, GPU call 29: kernel-trace stats of the bench commands with the final code (C3 incl. its sigma_omega = 0.1 companion, C5).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stats() { # name, bench args
  name=$1; shift
  timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o run -- python $R/bench.py "$@" > $O/$name.bench.log 2> $O/$name.err
  f=$(find $O/$name -name '*kernel_stats.csv' | head -1); cp $f $O/${name}_kernel_stats.csv; rm -rf $O/$name
  grep '"metric"' $O/$name.bench.log > $O/${name}_bench_line.json
}
stats c3 --workload c3 --steps 2 --warmup 1
stats c5 --workload c5 --steps 1 --warmup 1 --no-cpu-baseline
head -4 $O/c3_kernel_stats.csv | cut -c1-150
