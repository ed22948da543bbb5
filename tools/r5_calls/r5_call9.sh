# Round 5, GPU call 9: the streamed / pinned log_columns (tests that read logs through it), and the DEFAULT bench command end to end
# (other_workloads, materialise on the whole log, the CPU leg): its line and its wall time.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
grep -E "MemTotal|MemAvailable" /proc/meminfo > $O/host_meminfo.txt; nproc >> $O/host_meminfo.txt
T0=$(date +%s.%N); timeout 900 python bench.py > $O/c3_bench_line_call9.json 2> $O/bench9.err; echo "rc $? wall_s $(echo "$(date +%s.%N) - $T0" | bc)" > $O/bench_default_time.txt
python - <<'P'
import json,os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5'
try:
    d=json.loads(open(O+'/c3_bench_line_call9.json').read().strip().splitlines()[-1])
    print(json.dumps(dict(value=d['value'], ms=d['ms_per_step'], roofline={k:d['roofline'].get(k) for k in ('dominant','frac','ms')}, materialise=d['materialise'], others={k:(v['value'],v['ms_per_step'],v['dominant'],v['dominant_frac']) for k,v in (d['other_workloads'] or {}).items()}, drift=(d['sigma_omega_gt0'] or {}).get('value'), cpu=(d['cpu_baseline'] or {}).get('value')), indent=1), file=open(O+'/c3_bench_call9_summary.json','w'))
except Exception as e:
    print('ERR', e, file=open(O+'/c3_bench_call9_summary.json','w'))
P
