set -x
mkdir -p gpurun_out/r2_base
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python bench.py --workload c4shard --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2_base/c4shard.json 2> gpurun_out/r2_base/c4shard.err
timeout 120 python bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2_base/c2.json 2> gpurun_out/r2_base/c2.err
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r2_base/prof_c4 -- python bench.py --workload c4shard --users 300000 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r2_base/c4prof.json 2> gpurun_out/r2_base/c4prof.err
tail -c 600 gpurun_out/r2_base/c4shard.json; tail -c 400 gpurun_out/r2_base/c2.json
