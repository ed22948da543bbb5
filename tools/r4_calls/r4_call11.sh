# Round 4, GPU call 11: the whole GPU suite and the smoke test after the run-path options refactor (ABI v4), the default bench command.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests11.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests11.log; tail -6 $O/gpu_tests11.log | cut -c1-400
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > $O/c3_bench_line_default_flags.json 2> $O/c3_bench_default.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$O/c3_bench_line_default_flags.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['steps'], {k:v['ms'] for k,v in d['kernels'].items()}, d['roofline']['frac'])" | cut -c1-500
