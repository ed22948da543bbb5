# Round 3, GPU call 1: full -m gpu suite (incl. the new adversarial click / OUC tests and the bench shard test),
# full-size row-level parity of the default path vs the float64-only path, baseline bench lines, shard-size timings.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log; tail -5 $O/gpu_tests.log
timeout 900 python tools/full_scale_check.py c3 c2 c4shard > $O/full_scale_parity.txt 2> $O/full_scale_parity.err; echo "full_scale rc=$?"; tail -4 $O/full_scale_parity.txt
timeout 600 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$O/bench_c3.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()}, d['materialise'])"
timeout 600 python bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?"; python -c "
import json; d=json.loads(open('$O/bench_c5.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()}, d['roofline'])"
for u in 1250000 2500000 5000000; do
timeout 300 python bench.py --workload c3 --users $u --steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(workload='c3', users=$u, events_per_s=d['value'], ms_per_step=d['ms_per_step'], kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" >> $O/shard_sizes.jsonl
done
cat $O/shard_sizes.jsonl
