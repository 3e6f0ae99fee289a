#!/bin/bash
# Collect rocprofv3 kernel stats + PMC passes for bench.py on the GPU box.
# usage (via gpurun): bash tools/prof_pmc.sh <tag> [bench args...]
# PMC passes are separate runs, each with --kernel-trace only (pool rule).
set -u
TAG=${1:-run}; shift || true
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--steps 2 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -o s -- python3 bench.py $ARGS > $OUT/stats.log 2>&1
i=0
for PMC in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
  "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
  "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
  "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
  "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $R/$OUT/pmc$i -o p -- python3 bench.py $ARGS > $OUT/pmc$i.log 2>&1
done
python3 tools/prof_summary.py $OUT > $OUT/summary.md
cat $OUT/summary.md
