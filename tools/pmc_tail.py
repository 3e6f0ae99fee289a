#!/usr/bin/env python3
"""Per-launch counters of the walt:: mapping kernels of the LAST step of a rocprofv3 --pmc --kernel-trace run, in
launch order.  Usage: pmc_tail.py <dir with *counter_collection.csv> [min_value]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/*counter_collection.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "walt::" in r["Kernel_Name"]]
disp = collections.OrderedDict()
for r in rows:
    k = int(r["Dispatch_Id"])
    d = disp.setdefault(k, {"name": r["Kernel_Name"].replace("void walt::", "").split("(")[0], "c": {}})
    d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(disp)
last = max(i for i in ids if "k_ascii_to_2bit" in disp[i]["name"])
tot = collections.defaultdict(float)
for i in ids:
    if i < last:
        continue
    d = disp[i]
    if max(d["c"].values() or [0]) < float(sys.argv[2] if len(sys.argv) > 2 else 1000):
        continue
    print("%-44s %s" % (d["name"][:44], "  ".join("%s=%.4g" % (k.replace("_sum", ""), v) for k, v in sorted(d["c"].items()))))
    for k, v in d["c"].items():
        tot[k] += v
print("TOTAL " + "  ".join("%s=%.4g" % (k.replace("_sum", ""), v) for k, v in sorted(tot.items())))
