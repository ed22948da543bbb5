# Round 4, GPU call 23: the whole GPU suite on the final code (k_advance_run: a lane per user counts, a lane per event acts, no
# second event draw for events that cannot click), the default bench command, smoke, the sampled-oracle check at full size for C3
# with drift and C5 with fitted policies.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests23.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests23.log; tail -4 $O/gpu_tests23.log | cut -c1-400
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > $O/c3_bench_line_call23.json 2> $O/c3_bench23.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$O/c3_bench_line_call23.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()}, d['sigma_omega_gt0']['value'], d['cpu_baseline']['value'])" | cut -c1-400
rm -f $O/oracle_spot_check_full_size_call23.jsonl
timeout 900 python tests/oracle_spot_check.py c3drift c5trained --sample 2000 --out $O/oracle_spot_check_full_size_call23.jsonl > $O/spot23.log 2> $O/spot23.err; echo "spot rc=$?"
timeout 600 python tools/full_scale_check.py c3drift --users 2000000 > $O/full_scale_parity_rounds_c3drift.txt 2>&1; echo "full-scale c3drift rc=$?"; tail -1 $O/full_scale_parity_rounds_c3drift.txt | cut -c1-300
