# Round 4, GPU call 10: product slices in the float64 batch (k_exact_sums_h) — tests, shard sizes; the bench lines of the round
# (C3 default command, C2, C5, C5 with reference-fitted policies, the C4 shard); full-size row-level parity of C3 and C2 with the
# float64-only lock-step path (tools/full_scale_check.py).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "pipelined or last_round or memo_and or sum_cache_matches or sampled_oracle_parity_at_bench_size and c3" > $O/gpu_tests10.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests10.log; tail -4 $O/gpu_tests10.log | cut -c1-300
rm -f $O/c3_shard_sizes_sliced.jsonl
for u in 1250000 2500000 10000000; do
timeout 300 python bench.py --workload c3 --users $u --steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(workload='c3', users=$u, events_per_s=d['value'], ms_per_step=d['ms_per_step'], kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" >> $O/c3_shard_sizes_sliced.jsonl
done
cat $O/c3_shard_sizes_sliced.jsonl
timeout 900 python bench.py --steps 10 --warmup 3 > $O/c3_bench_line_final.json 2> $O/c3_bench_final.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$O/c3_bench_line_final.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()}, d['roofline']['frac'], d['roofline'].get('traffic_bytes_per_unit'), d['sigma_omega_gt0']['value'], d['cpu_baseline']['value'], d['cpu_baseline']['reference_numpy'].get('estimated_on_this_box'))" | cut -c1-900
for wl in c2 c5 c5trained c4shard; do
timeout 600 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line > $O/${wl}_bench_line_final.json 2> $O/${wl}_bench_final.err; python -c "
import json; d=json.loads(open('$O/${wl}_bench_line_final.json').read().strip().splitlines()[-1]); print('$wl', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})" | cut -c1-600
done
timeout 900 python tools/full_scale_check.py c3 c2 > $O/full_scale_parity_final_c3_c2.txt 2> $O/full_scale_final.err; echo "full_scale rc=$?"; tail -3 $O/full_scale_parity_final_c3_c2.txt | cut -c1-300
