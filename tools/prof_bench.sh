#!/bin/bash
# rocprofv3 kernel-trace summary of one bench.py invocation; results under gpurun_out/<name>/
# usage: tools/prof_bench.sh <name> [bench.py args...]
NAME=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/bench.log 2>&1
grep '"metric"' $OUT/bench.log
find $OUT -name '*kernel_stats.csv' | head -1 | xargs head -15
