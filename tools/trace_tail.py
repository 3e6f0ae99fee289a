#!/usr/bin/env python3
"""Durations of the walt:: kernels of the LAST step of a rocprofv3 --kernel-trace run, in launch order (launches
shorter than 20 us are left out).  Usage: trace_tail.py <dir with *kernel_trace.csv>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "walt::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "k_ascii_to_2bit" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d >= 0.02:
        print("%8.3f ms  +%8.3f  %s" % (d, (int(r["Start_Timestamp"]) - t0) / 1e6, r["Kernel_Name"].replace("void walt::", "")[:60]))
