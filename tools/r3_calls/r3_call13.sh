# Round 3, GPU call 13: the certificate from both ends of the chunk (cert_two_sided) in k_walk2 / k_walk_solo; does the
# float64 MFMA co-execute with float64 VALU work (tools/ubench/mfma_f64_coexec.hip).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "certificate or sum_cache or memo or wave_per_user or every_K or sigma_omega_zero" > $O/gpu_tests13.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests13.log; tail -5 $O/gpu_tests13.log | cut -c1-400
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py $B $WL 2>$O/ab13_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(json.dumps(dict(name='$name', events_per_s=round(d['value']/1e6,1), ms_per_step=round(d['ms_per_step'],2), exact_fraction=r.get('exact_fraction'), round1_ms=r.get('round1_ms'), later_rounds_ms=r.get('later_rounds_ms'), kernels={k:(v['ms']) for k,v in d['kernels'].items()})))" >> $O/ab13.jsonl
}
rm -f $O/ab13.jsonl
B="--steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise"
for WLN in c3 c2; do WL="--workload $WLN"; run ${WLN} A=1; done
WL="--workload c3 --users 1250000"; run c3_shard_1250000 A=1
cat $O/ab13.jsonl
timeout 600 python tools/full_scale_check.py c3 c2 > $O/full_scale_13.txt 2>&1; echo "full scale rc=$?"; grep verdict $O/full_scale_13.txt
timeout 120 tools/ubench/mfma_f64_coexec.bin > $O/ubench_mfma_f64_coexec.txt 2>&1; cat $O/ubench_mfma_f64_coexec.txt
