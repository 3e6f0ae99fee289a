#!/bin/bash
# Where the single-end step goes, launch by launch (GPU box): kernel trace of the last step, HBM read requests and
# written bytes per launch.  bash tools/prof_stage.sh <tag> [bench args ...]   (one stream: --opt se_pipe=0)
set -u
TAG=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
OUT=gpurun_out/stage_$TAG; mkdir -p $OUT
ARGS="--no-extra --no-cpu-baseline --steps 2 --warmup 1 $*"
rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/trace -o t -- python3 bench.py $ARGS > $OUT/trace.json 2> $OUT/trace.log || { tail -5 $OUT/trace.log; exit 1; }
grep -E "staged rounds|heavy pass:" $OUT/trace.log
python3 tools/trace_tail.py $OUT/trace > $OUT/trace_tail.txt; cat $OUT/trace_tail.txt
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $R/$OUT/pmc_rd -o p -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_rd.log || { tail -5 $OUT/pmc_rd.log; exit 1; }
python3 tools/pmc_tail.py $OUT/pmc_rd 100000 > $OUT/pmc_rd.txt; cat $OUT/pmc_rd.txt
rocprofv3 --pmc WRITE_SIZE TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $R/$OUT/pmc_wr -o p -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_wr.log || { tail -5 $OUT/pmc_wr.log; exit 1; }
python3 tools/pmc_tail.py $OUT/pmc_wr 1000 > $OUT/pmc_wr.txt; cat $OUT/pmc_wr.txt
rm -rf $OUT/trace $OUT/pmc_rd $OUT/pmc_wr
