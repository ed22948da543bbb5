# Round 5, GPU call 33: which helper events count, from two 64-bit masks over the dealt events (LDS atomic OR, then wave-uniform mask
# arithmetic) instead of a flag byte per event read back along every chain — against the build before it on one box; parity.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
rm -f $O/ab_call33.jsonl
run() {  # name, workload, env...
  name=$1; wl=$2; shift; shift
  env "$@" timeout 150 python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab33.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events=d['config'].get('events_per_step'), ctr=d['config'].get('ctr'), ms_per_step=round(d['ms_per_step'],2), kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items()})))" >> $O/ab_call33.jsonl
}
L=$R/recogym_amd/csrc/librecogym_hip_prev.so
for i in 1 2 3; do
run chain_masks c3
run before c3 RECOGYM_HIP_LIB=$L
done
run chain_masks c2
run before c2 RECOGYM_HIP_LIB=$L
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "walk or memo or sum_cache or hip_matches_oracle or organic_only or phantom or ouc or history or long_runs" 2>&1 | tail -3 > $O/gpu_tests_call33.txt
