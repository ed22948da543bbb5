# Round 4, GPU call 9 (final code): the whole GPU suite; kernel-trace stats + PMC passes of the benched 10 M-user C3 run
# (tools/r4_profiles.sh); what a rank's share of a strongly scaled C3 costs on one GPU.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests9.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests9.log; tail -6 $O/gpu_tests9.log | cut -c1-400
bash tools/r4_profiles.sh > $O/r4_profiles.log 2>&1; tail -5 $O/r4_profiles.log
cd $R
rm -f $O/c3_shard_sizes.jsonl
for u in 1250000 2500000 5000000 10000000; do
timeout 300 python bench.py --workload c3 --users $u --steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(workload='c3', users=$u, events_per_s=d['value'], ms_per_step=d['ms_per_step'], kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" >> $O/c3_shard_sizes.jsonl
done
cat $O/c3_shard_sizes.jsonl
