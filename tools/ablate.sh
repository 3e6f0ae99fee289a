#!/bin/bash
# attribute HBM read requests of k_map_se to its phases (diagnostic)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for AB in 0 1 2 4; do
  export WALT_AMD_ABLATE=$AB
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/ablate$AB -o a -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/ablate$AB.log 2>&1
  python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/ablate$AB/a_counter_collection.csv")) if "k_map_se<8>" in r["Kernel_Name"]]
last={}
for r in rows: last[r["Counter_Name"]]=(float(r["Counter_Value"]), (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
print("ablate=$AB", {k:(round(v[0]/50e6,2), round(v[1],2)) for k,v in last.items()}, "(per read, kernel ms)")
PY
done
