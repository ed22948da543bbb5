# Round 4, GPU call 27: the default build with two product tiles in the sweep's LDS ring (RG_SWEEP_NB=2): the GPU suite without its
# two slowest groups (the 10^4-class LogReg oracle cases and the sampled-oracle runs: neither touches the ring), smoke, C3.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "not frozen_logreg and not sampled_oracle" > $O/gpu_tests27.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests27.log; tail -4 $O/gpu_tests27.log | cut -c1-400
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | cut -c1-200
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-materialise > $O/c3_bench_line_call27.json 2> $O/c3_bench27.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$O/c3_bench_line_call27.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()}, d['sigma_omega_gt0']['value'])" | cut -c1-400
