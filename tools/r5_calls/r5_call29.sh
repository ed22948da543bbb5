# Round 5, GPU call 29: evidence on the final code (after the walk work of calls 17-28) — the whole GPU suite, smoke, the default
# bench command, kernel-trace stats + PMC passes of the benched C3 run (tools/r5_profiles.sh), the sampled-oracle check at full
# size, every row against the float64-only path at full size (C3, C2), shard sizes.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests_call29_full.txt 2>&1; echo "pytest rc=$?" >> $O/gpu_tests_call29_full.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke_call29.txt 2>&1
timeout 600 python bench.py > $O/c3_bench_line_call29.json 2> $O/c3_bench29.err; echo "bench rc=$?" >> $O/smoke_call29.txt
bash tools/r5_profiles.sh > $O/profiles29.log 2>&1
rm -f $O/oracle_spot_check_full_size.jsonl
timeout 900 python tests/oracle_spot_check.py c3 c2 --sample 2000 --out $O/oracle_spot_check_full_size.jsonl > $O/spot29.log 2> $O/spot29.err; echo "spot rc=$?" >> $O/smoke_call29.txt
timeout 900 python tools/full_scale_check.py c3 c2 > $O/full_scale_parity_c3_c2.txt 2>&1; echo "full-scale rc=$?" >> $O/smoke_call29.txt
rm -f $O/c3_shard_sizes.jsonl
for u in 1250000 2500000 5000000; do
timeout 120 python bench.py --workload c3 --users $u --steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(workload='c3', users=$u, events_per_s=round(d['value']/1e6,1), ms_per_step=round(d['ms_per_step'],2), kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" >> $O/c3_shard_sizes.jsonl
done
