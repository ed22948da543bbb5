mkdir -p gpurun_out/r2_t28
run() { # tag env...
tag=$1; shift
env "$@" timeout 300 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --no-drift-line > gpurun_out/r2_t28/c3_$tag.json 2> gpurun_out/r2_t28/c3.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_t28/c3_$tag.json').read().strip().splitlines()[-1]); w=d['kernels']['walk']; print('$tag', round(d['value']/1e6,1), round(d['ms_per_step'],1), w['ms'], w['round1_ms'], w['round2_ms'])
PY
}
run r8 X=1
run r1 RECOGYM_WALK_REFILL=1
run r3 RECOGYM_WALK_REFILL=3
run r16 RECOGYM_WALK_REFILL=16
