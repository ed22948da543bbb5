set -x
mkdir -p gpurun_out/r2_t2
python -m pytest tests -m gpu -q -x -k "sigma_omega_zero or sum_cache or certificate_is_sound or fixture" 2>&1 | tail -30
timeout 200 python bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2_t2/c3.json 2> gpurun_out/r2_t2/c3.err; tail -c 1500 gpurun_out/r2_t2/c3.json; tail -5 gpurun_out/r2_t2/c3.err
timeout 100 python bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2_t2/c2.json 2> gpurun_out/r2_t2/c2.err; tail -c 700 gpurun_out/r2_t2/c2.json
