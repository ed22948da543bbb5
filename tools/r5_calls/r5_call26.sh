# Round 5, GPU call 26: the draw's chunk recomputed in plain fp32 again, with its OWN budget delta_c in the certificate (the sweep
# terms keep delta ~ 1e-5): parity (adversarial certificate tests), C3 / C2, every row of C3 against the float64-only path.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "walk or memo or sum_cache or hip_matches_oracle or organic_only or phantom or ouc or history or certificate or error_free" 2>&1 | tail -4 > $O/gpu_tests_call26.txt
rm -f $O/ab_call26.jsonl
run() {  # name, workload, env...
  name=$1; wl=$2; shift; shift
  env "$@" timeout 150 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab26.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events=d['config'].get('events_per_step'), ctr=d['config'].get('ctr'), ms_per_step=round(d['ms_per_step'],2), kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items()})))" >> $O/ab_call26.jsonl
}
run c3 c3
run c3 c3
run c2 c2
timeout 600 python tools/full_scale_check.py c3 > $O/full_scale_parity_call26_c3.txt 2>>$O/ab26.err
tail -3 $O/full_scale_parity_call26_c3.txt
