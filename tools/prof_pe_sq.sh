#!/bin/bash
# SQ counters of the paired-end kernels, serialized (GPU box): bash tools/prof_pe_sq.sh <tag>
set -u
TAG=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
OUT=gpurun_out/pesq_$TAG; mkdir -p $OUT
ARGS="--mode pe --no-extra --no-cpu-baseline --steps 1 --warmup 1 --opt pe_serial=1 $*"
i=0
for PMC in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $R/$OUT/p$i -o p -- python3 bench.py $ARGS > /dev/null 2> $OUT/p$i.log || { tail -5 $OUT/p$i.log; exit 1; }
done
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for sub in ("p1", "p2"):
    for f in glob.glob(out + "/" + sub + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "walt::k_pe" not in k and "k_pe_" not in k: continue
            nm = k.split("(")[0].replace("void ", "").replace("walt::", "")
            tot[nm][r["Counter_Name"]] += float(r["Counter_Value"])
for nm, c in sorted(tot.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    g = c.get("GRBM_GUI_ACTIVE", 0) / 8  # cycles (8 XCDs)
    if g <= 0: continue
    simd_q = 1024 * g / 4  # SIMD quad-cycles available
    print("%-26s ms %7.2f  waves %9.0f  VALU/wave %7.0f  LDS/wave %6.0f  VMEM_RD/wave %6.0f  valu_util %.2f  lds_active %.2f  any_active/wave-cycles %.2f  wait/wave-cycles %.2f  wait_lds/wave-cycles %.2f" % (
        nm[:26], g / 2.4e6, c.get("SQ_WAVES", 0), c.get("SQ_INSTS_VALU", 0) / max(1, c.get("SQ_WAVES", 1)),
        c.get("SQ_INSTS_LDS", 0) / max(1, c.get("SQ_WAVES", 1)), c.get("SQ_INSTS_VMEM_RD", 0) / max(1, c.get("SQ_WAVES", 1)),
        c.get("SQ_ACTIVE_INST_VALU", 0) / simd_q, c.get("SQ_ACTIVE_INST_LDS", 0) / simd_q,
        c.get("SQ_ACTIVE_INST_ANY", 0) / max(1, c.get("SQ_WAVE_CYCLES", 1)), c.get("SQ_WAIT_ANY", 0) / max(1, c.get("SQ_WAVE_CYCLES", 1)),
        c.get("SQ_WAIT_INST_LDS", 0) / max(1, c.get("SQ_WAVE_CYCLES", 1))))
PY
rm -rf $OUT/p1 $OUT/p2
