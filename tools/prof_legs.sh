#!/bin/bash
# Kernel statistics and HBM traffic of every bench leg, one leg per bench.py command so that the counters of a leg
# are not mixed with another's (GPU box; run through gpurun).  Per leg: one rocprofv3 --kernel-trace --stats run
# and two PMC runs (read requests by size; WRITE_SIZE) -- separate passes, each with --kernel-trace only, as the
# pool's rule and the TCC slot budget require.  tools/prof_traffic.py turns them into profiles/<tag>_<leg>_*.{csv,md}
# and the entries of profiles/traffic.json that bench.py's roofline.traffic reads.
#   bash tools/prof_legs.sh <tag> [legs...]      legs: se100 se150 se150ag pe100 pe150 (default: all)
set -u
TAG=${1:-round3}; shift || true
LEGS=${*:-se100 se150 se150ag pe100 pe150}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
COMMON="--steps 3 --warmup 1 --no-cpu-baseline --no-extra"
for LEG in $LEGS; do
  case $LEG in
    se100)   ARGS="$COMMON" ;;
    se150)   ARGS="$COMMON --read-len 150 --max-mismatches 10 --reads 25000000" ;;
    se150ag) ARGS="$COMMON --read-len 150 --max-mismatches 10 --reads 25000000 --ag" ;;
    pe100)   ARGS="$COMMON --mode pe" ;;
    pe150)   ARGS="$COMMON --mode pe --read-len 150 --max-mismatches 10 --reads 25000000 --pbat" ;;
    *) echo "unknown leg $LEG"; exit 2 ;;
  esac
  OUT=gpurun_out/prof_${TAG}_$LEG
  mkdir -p $OUT
  echo "== $LEG: python3 bench.py $ARGS"
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -o s -- python3 bench.py $ARGS > $OUT/bench.json 2> $OUT/stats.log || { echo "stats run failed"; tail -n 5 $OUT/stats.log; exit 1; }
  rm -f $OUT/stats/s_kernel_trace.csv
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $R/$OUT/pmc_rd -o p -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_rd.log || { echo "pmc read pass failed"; tail -n 5 $OUT/pmc_rd.log; exit 1; }
  rm -f $OUT/pmc_rd/p_kernel_trace.csv
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_wr -o p -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_wr.log || { echo "pmc write pass failed"; tail -n 5 $OUT/pmc_wr.log; exit 1; }
  rm -f $OUT/pmc_wr/p_kernel_trace.csv
  rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum --kernel-trace --output-format csv -d $R/$OUT/pmc_tlb -o p -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_tlb.log || echo "tlb pass failed (kept going)"
  rm -f $OUT/pmc_tlb/p_kernel_trace.csv
  python3 tools/prof_traffic.py $OUT $LEG $TAG
  # the raw per-dispatch counter files (every torch kernel of the workload generator included) run to tens of MB:
  # gpurun brings back at most 64 MiB, and summary.md / kernel_stats.csv / traffic_entry.json are what is kept
  rm -rf $OUT/pmc_rd $OUT/pmc_wr $OUT/pmc_tlb $OUT/stats
done
