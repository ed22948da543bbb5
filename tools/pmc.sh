#!/bin/bash
# usage: tools/pmc.sh <name> "<counters>" <cmd...>   -> gpurun_out/<name>/
NAME=$1; shift; CNT=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT -o run -- "$@" > $OUT/cmd.log 2>&1
tail -3 $OUT/cmd.log
ls $OUT
