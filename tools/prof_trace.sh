#!/bin/bash
# Kernel trace of the last single-end step, launch by launch (GPU box).  bash tools/prof_trace.sh <tag> [bench args ...]
set -u
TAG=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
OUT=gpurun_out/trace_$TAG; mkdir -p $OUT
ARGS="--no-extra --no-cpu-baseline --steps 2 --warmup 1 $*"
rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/trace -o t -- python3 bench.py $ARGS > $OUT/trace.json 2> $OUT/trace.log || { tail -5 $OUT/trace.log; exit 1; }
grep -E "staged rounds|heavy pass:" $OUT/trace.log
python3 tools/trace_tail.py $OUT/trace > $OUT/trace_tail.txt; cat $OUT/trace_tail.txt
rm -rf $OUT/trace
