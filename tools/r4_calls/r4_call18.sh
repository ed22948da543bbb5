# Round 4, GPU call 18: kernel-trace stats of the rounds (C3 with drift at 4 M users, C5 at its bench size) — which of
# k_advance_run / k_drift / k_round_rows / the acts the "advance" slot is made of; the LogReg overflow-path parity case.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "frozen_logreg" > $O/gpu_tests18.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests18.log; tail -4 $O/gpu_tests18.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
stats() { # name, bench args
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o run -- python $R/bench.py "$@" > $O/$name.bench.log 2> $O/$name.err
  f=$(find $O/$name -name '*kernel_stats.csv' | head -1); cp $f $O/${name}_kernel_stats.csv; rm -rf $O/$name
  grep '"metric"' $O/$name.bench.log > $O/${name}_bench_line.json
  head -14 $O/${name}_kernel_stats.csv | cut -c1-200
}
stats c3drift_rounds --workload c3drift --users 4000000 --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise
stats c5_rounds --workload c5 --steps 1 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise
