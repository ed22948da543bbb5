# Round 5, GPU call 24: the search iteration's diet — the lanes a chunk pass serves from a list in LDS (was: the next eight set bits of
# the ballot, one by one, ~75 instructions per pass), the prefix scans as compare / add / select (prefixes ascend: no maximum needed).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "walk or memo or sum_cache or hip_matches_oracle or organic_only or phantom or ouc or history" 2>&1 | tail -4 > $O/gpu_tests_call24.txt
rm -f $O/ab_call24_diet.jsonl
run() {  # name, workload, env...
  name=$1; wl=$2; shift; shift
  env "$@" timeout 150 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab24.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events=d['config'].get('events_per_step'), ctr=d['config'].get('ctr'), ms_per_step=round(d['ms_per_step'],2), kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items()})))" >> $O/ab_call24_diet.jsonl
}
run c3 c3
run c3 c3
run c2 c2
L=$R/recogym_amd/csrc/librecogym_hip_walktiming.so
RECOGYM_HIP_LIB=$L timeout 200 python tools/walk_kinds.py c3 2>>$O/ab24.err | tail -1 > $O/walk_kinds_call24.jsonl
