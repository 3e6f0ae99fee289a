#!/usr/bin/env python3
"""Per-kernel totals of the walt:: kernels of a rocprofv3 --kernel-trace run, per bench step (launches / steps given).
Usage: trace_sum.py <dir with *kernel_trace.csv> <steps incl. warmup>"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
steps = int(sys.argv[2])
tot = collections.defaultdict(float); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "walt::" not in k or any(x in k for x in ("k_count_buckets", "k_keys", "k_dir_", "k_make_ent", "k_check_", "k_win_", "k_mark", "k_strand", "k_fill", "k_unpack", "k_ent_pos")):
        continue
    nm = k.split("(")[0].replace("void ", "").replace("walt::", "")
    tot[nm] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    cnt[nm] += 1
for nm in sorted(tot, key=lambda n: -tot[n]):
    print("%10.2f ms/step  %6.1f launches/step  %s" % (tot[nm] / steps, cnt[nm] / steps, nm))
