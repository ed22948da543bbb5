# Round 4, GPU call 15: the whole GPU suite after retiring the superseded kernel variants; the default bench command.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests15.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests15.log; tail -6 $O/gpu_tests15.log | cut -c1-400
timeout 600 python bench.py --no-cpu-baseline --no-drift-line > $O/c3_bench_line_call15.json 2> $O/c3_bench15.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$O/c3_bench_line_call15.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})" | cut -c1-400
