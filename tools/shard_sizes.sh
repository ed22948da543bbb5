# What a rank's share of a strongly scaled C3 costs on one GPU: 1.25 / 2.5 / 5 M users.
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/r3
mkdir -p $O
rm -f $O/c3_shard_sizes.jsonl
for u in 1250000 2500000 5000000 10000000; do
timeout 120 python bench.py --workload c3 --users $u --steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(workload='c3', users=$u, events_per_s=round(d['value']/1e6,1), ms_per_step=round(d['ms_per_step'],2), kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" | tee -a $O/c3_shard_sizes.jsonl
done
