# Round 3, GPU call 12: the certificate on correlated errors (cert_correlated).  Soundness tests first, then what it buys
# (C3 / C2 / c3drift / C5 / C4 shard), then row-level parity at full size against the float64-only path.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "certificate or sum_cache or memo or wave_per_user or every_K or sigma_omega_zero" > $O/gpu_tests12.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests12.log; tail -5 $O/gpu_tests12.log | cut -c1-400
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py $B $WL 2>$O/ab12_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events_per_s=round(d['value']/1e6,1), ms_per_step=round(d['ms_per_step'],2), exact_fraction=d['roofline'].get('exact_fraction'), kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" >> $O/ab12.jsonl
}
rm -f $O/ab12.jsonl
B="--steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
for WLN in c3 c2; do WL="--workload $WLN"; run ${WLN} A=1; done
B="--steps 1 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
for WLN in c3drift c5 c4shard; do WL="--workload $WLN"; run ${WLN} A=1; done
cat $O/ab12.jsonl
timeout 600 python tools/full_scale_check.py c3 c2 > $O/full_scale_12.txt 2>&1; echo "full scale rc=$?"; grep verdict $O/full_scale_12.txt
