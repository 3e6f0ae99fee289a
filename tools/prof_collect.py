#!/usr/bin/env python3
"""File the per-leg profiles that tools/prof_legs.sh left under gpurun_out/prof_<tag>_<leg>/ into profiles/:
   profiles/<tag>_<leg>_summary.md, profiles/<tag>_<leg>_kernel_stats.csv, profiles/traffic.json[<leg>].
   python3 tools/prof_collect.py <tag>"""
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tj = os.path.join(root, "profiles", "traffic.json")
try:
    allt = json.load(open(tj))
    if "hbm_bytes_per_launch" in allt:  # round-1 layout (one workload)
        allt = {}
except (OSError, ValueError):
    allt = {}
for d in sorted(glob.glob(os.path.join(root, "gpurun_out", "prof_%s_*" % tag))):
    leg = os.path.basename(d)[len("prof_%s_" % tag):]
    if not os.path.exists(os.path.join(d, "traffic_entry.json")):
        print("skip", d)
        continue
    shutil.copy(os.path.join(d, "summary.md"), os.path.join(root, "profiles", "%s_%s_summary.md" % (tag, leg)))
    if os.path.exists(os.path.join(d, "kernel_stats.csv")):
        shutil.copy(os.path.join(d, "kernel_stats.csv"), os.path.join(root, "profiles", "%s_%s_kernel_stats.csv" % (tag, leg)))
    allt[leg] = json.load(open(os.path.join(d, "traffic_entry.json")))
    print("filed", leg, "%.1f GB per step" % ((allt[leg].get("hbm_bytes_per_launch") or allt[leg].get("hbm_bytes_per_step")) / 1e9))
json.dump(allt, open(tj, "w"), indent=1, sort_keys=True)
