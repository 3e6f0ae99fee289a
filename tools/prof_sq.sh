#!/bin/bash
# SQ-level counters of every mapping launch of the last single-end step (GPU box): is a kernel bound by instruction
# issue or by waiting, and how many of its lanes work (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU: lanes per
# VALU instruction)?  bash tools/prof_sq.sh <tag> [bench args...]
set -u
TAG=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
OUT=gpurun_out/sq_$TAG; mkdir -p $OUT
ARGS="--no-extra --no-cpu-baseline --steps 2 --warmup 1 $*"
i=0
for PMC in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
  "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $R/$OUT/p$i -o p -- python3 bench.py $ARGS > /dev/null 2> $OUT/p$i.log || { tail -5 $OUT/p$i.log; exit 1; }
  python3 tools/pmc_tail.py $OUT/p$i 1000 > $OUT/p$i.txt; cat $OUT/p$i.txt
  rm -rf $OUT/p$i
done
