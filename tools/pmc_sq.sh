#!/bin/bash
# SQ-level counters of k_map_se for a given WALT_AMD_ABLATE value (diagnostic)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
AB=${1:-0}; export WALT_AMD_ABLATE=$AB
i=0
for PMC in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
  "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
  "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT32"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $R/gpurun_out/sq${AB}_$i -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/sq${AB}_$i.log 2>&1
done
python3 - <<PY
import csv, glob
vals={}
for f in sorted(glob.glob("gpurun_out/sq${AB}_*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "k_map_se<8>" in r["Kernel_Name"]:
            vals[r["Counter_Name"]]=(float(r["Counter_Value"]), (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
print("ablate=$AB")
for k,v in vals.items(): print("  %-28s %.4g   (kernel %.2f ms)" % (k, v[0], v[1]))
PY
