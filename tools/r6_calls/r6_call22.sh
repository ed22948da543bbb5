# Round 6, GPU call 22: final-tree evidence — the whole GPU suite, smoke, the default bench command; C5 with the reference-fitted
# policies (P = 100: one product tile, k_draw_bf16p again) and c3drift / C5 lines.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests/ -q -m gpu 2>&1 | tail -8 > $O/gpu_tests_call22.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_call22.txt 2>&1
run() { # name, env, args
  name=$1; envs=$2; shift; shift
  env $envs timeout 600 python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab22.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events=d['config']['events_per_step'], ctr=d['config']['ctr'], ms_per_step=round(d['ms_per_step'],2), value=d['value'], kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items()})))" | tee -a $O/ab_call22_lines.jsonl
}
rm -f $O/ab_call22_lines.jsonl
run c5trained "X=1" --workload c5trained
run c5 "X=1" --workload c5
run c3drift "X=1" --workload c3drift
timeout 900 python bench.py > $O/bench_default_call22.json 2> $O/bench_default_call22.err
tail -c 600 $O/bench_default_call22.json
