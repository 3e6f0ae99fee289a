#!/bin/bash
# Paired-end kernels one at a time (option pe_serial = 1): per-kernel time per step (kernel trace only).
#   bash tools/prof_pe_trace.sh <tag> [bench args ...]
set -u
TAG=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
OUT=gpurun_out/pet_$TAG; mkdir -p $OUT
ARGS="--mode pe --no-extra --no-cpu-baseline --steps 2 --warmup 1 --opt pe_serial=1 $*"
rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/trace -o t -- python3 bench.py $ARGS > $OUT/trace.json 2> $OUT/trace.log || { tail -5 $OUT/trace.log; exit 1; }
python3 tools/trace_sum.py $OUT/trace 3 > $OUT/trace_sum.txt; cat $OUT/trace_sum.txt
grep -E "ms/step|lists" $OUT/trace.log | tail -3
rm -rf $OUT/trace
