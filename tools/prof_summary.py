#!/usr/bin/env python3
"""Summarise tools/prof_pmc.sh output: per-kernel average duration and the last
launch's PMC values for the walt kernels."""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]
print("# rocprofv3 summary: %s\n" % out)
st = glob.glob(os.path.join(out, "stats", "*kernel_stats.csv"))
if st:
    print("| kernel | calls | avg ms | min ms | max ms |\n|---|---|---|---|---|")
    for r in csv.DictReader(open(st[0])):
        if "walt::" in r["Name"]:
            nm = r["Name"].split("(")[0].replace("void ", "")
            print("| %s | %s | %.3f | %.3f | %.3f |" % (nm, r["Calls"], float(r["AverageNs"]) / 1e6,
                                                      float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
print("\nPMC (value of the LAST launch of each kernel in its pass):\n")
vals = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(out, "pmc*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "walt::k_map" not in k and "walt::k_pack" not in k and "walt::k_pe" not in k:
            continue
        nm = k.split("(")[0].replace("void ", "")
        vals.setdefault(nm, collections.OrderedDict())[r["Counter_Name"]] = float(r["Counter_Value"])
for nm, d in vals.items():
    print("## %s\n" % nm)
    for c, v in d.items():
        print("- %s = %.6g" % (c, v))
    print()

# HBM traffic per launch of the mapping kernels (pass 1 + literal pass): every
# TCC_EA0_RDREQ is a 128-byte request on this kernel (TCC_EA0_RDREQ_128B == RDREQ);
# WRITE_SIZE is in KB.  Written for bench.py's roofline.traffic.
import json
rd = sum(d.get("TCC_EA0_RDREQ_sum", 0) for nm, d in vals.items() if "k_map_se" in nm)
wr = sum(d.get("WRITE_SIZE", 0) for nm, d in vals.items() if "k_map_se" in nm)
if rd:
    meta = {"hbm_bytes_per_launch": rd * 128 + wr * 1024, "read_requests_128B": rd, "write_kb": wr,
            "kernels": [nm for nm in vals if "k_map_se" in nm], "reads_per_launch": 50000000, "genome_bp": 3095677412,
            "source": out}
    # keep the per-read algorithmic counters bench.py's N>1 runs rely on (profiles/traffic.json)
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    try:
        old = json.load(open(here))
        if "algorithmic_per_read" in old:
            meta["algorithmic_per_read"] = old["algorithmic_per_read"]
    except (OSError, ValueError):
        pass
    with open(os.path.join(out, "traffic.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("traffic.json:", json.dumps(meta))
