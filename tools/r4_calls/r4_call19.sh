# Round 4, GPU call 19: the evidence with the rounds in (k_advance_run) — the whole GPU suite, the bench lines of every workload, the
# sampled-oracle check at full bench sizes for the workloads that run in rounds, rounds against float64 lock-step on every row.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests19.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests19.log; tail -6 $O/gpu_tests19.log | cut -c1-400
timeout 600 python bench.py > $O/c3_bench_line_rounds.json 2> $O/c3_bench19.err; echo "bench rc=$?"
for wl in c3drift c5 c5trained c4shard c2; do
  timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-drift-line > $O/${wl}_bench_line_rounds.json 2> $O/${wl}_bench19.err; echo "$wl rc=$?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r4/*_bench_line_rounds.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value'] / 1e6, 1), 'M ev/s', round(d['ms_per_step'], 2), 'ms', {k: v['ms'] for k, v in d.get('kernels', {}).items()}, (d.get('sigma_omega_gt0') or {}).get('value'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
rm -f $O/oracle_spot_check_full_size_rounds.jsonl
timeout 900 python tests/oracle_spot_check.py c3drift c4shard c5trained --sample 2000 --out $O/oracle_spot_check_full_size_rounds.jsonl > $O/spot19.log 2> $O/spot19.err; echo "spot rc=$?"; tail -2 $O/spot19.err
timeout 900 python tools/full_scale_check.py c3drift --users 2000000 > $O/full_scale_parity_rounds_c3drift.txt 2>&1; echo "full-scale c3drift rc=$?"; tail -1 $O/full_scale_parity_rounds_c3drift.txt | cut -c1-300
timeout 900 python tools/full_scale_check.py c5 c4shard > $O/full_scale_parity_rounds_c5_c4.txt 2>&1; echo "full-scale c5 c4shard rc=$?"; grep verdict $O/full_scale_parity_rounds_c5_c4.txt | cut -c1-300
