# Round 6, GPU call 24: tile DMA spread over the MFMA stream's issue slots (k_draw_tp, k_draw_tpw) against the burst behind the tile
# barrier (-DRG_TP_DMA_SPREAD=0): parity subset on the default build, then c3drift / C5 / C4 lines and the step-0 probes of both.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_hip_parity.py -q -m gpu -k "lds_search or certif or run_ahead or wide_logit" 2>&1 | tail -5 > $O/gpu_tests_call24.txt
run() { # name, env, args
  name=$1; envs=$2; shift; shift
  env $envs timeout 600 python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab24.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='$name', events=d['config']['events_per_step'], ctr=d['config']['ctr'], ms_per_step=round(d['ms_per_step'],2), value=d['value'], kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items()})))" | tee -a $O/ab_call24_lines.jsonl
}
rm -f $O/ab_call24_lines.jsonl $O/ab24.err
NS=RECOGYM_HIP_LIB=$R/recogym_amd/csrc/librecogym_hip_nospread.so
run c3drift_spread "X=1" --workload c3drift
run c3drift_burst "$NS" --workload c3drift
run c4_spread "X=1" --workload c4shard
run c4_burst "$NS" --workload c4shard
run c5_spread "X=1" --workload c5
run c5_burst "$NS" --workload c5
run c3drift_spread2 "X=1" --workload c3drift
run c3drift_burst2 "$NS" --workload c3drift
