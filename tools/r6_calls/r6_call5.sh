# Round 6, GPU call 5: the > 2^31 raw-row regression test (and the same test on a build with round 5's bug put back: must FAIL);
# a kernel trace of one c3drift run, per dispatch (which rounds carry k_pick's / k_draw_tp's time).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "raw_log_rows_beyond" 2>&1 | tail -8 > $O/gpu_tests_call5_rowbase.txt
RECOGYM_HIP_LIB=$R/recogym_amd/csrc/librecogym_hip_signed_rfl.so timeout 1200 python -m pytest tests/test_hip_parity.py -q -m gpu -k "raw_log_rows_beyond" 2>&1 | tail -14 > $O/gpu_tests_call5_rowbase_with_the_bug_put_back.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3drift -o run -- python $R/bench.py --workload c3drift --single-run --steps 1 --warmup 0 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads > $O/trace_c3drift.bench.log 2> $O/trace_c3drift.err
f=$(find $O/trace_c3drift -name '*kernel_stats.csv' | head -1); cp $f $O/c3drift_kernel_stats_call5.csv
f=$(find $O/trace_c3drift -name '*kernel_trace.csv' | head -1)
python $R/tools/trace_rounds.py $f $O/c3drift_rounds_call5.json k_pick k_draw_tp k_draw_bf16p k_advance_run k_drift k_round_rows k_exact > $O/c3drift_rounds_call5.txt 2>&1
rm -rf $O/trace_c3drift
