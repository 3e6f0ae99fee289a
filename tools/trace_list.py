#!/usr/bin/env python3
"""Durations of every launch of the kernels whose name contains <pattern>, in launch order (last <n> launches).
Usage: trace_list.py <dir with *kernel_trace.csv> <pattern> [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
pat = sys.argv[2]; n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = [r for r in csv.DictReader(open(f)) if "walt::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if pat in r["Kernel_Name"]][-n:]
print(" ".join("%.2f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in sel))
