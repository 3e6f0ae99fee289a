#!/usr/bin/env python3
"""HBM traffic of one paired-end step from the PMC passes of `tools/prof_pmc.sh <tag> --mode pe`:
sums TCC_EA0_RDREQ (x 128 B; all requests of these kernels are 128-byte ones) and WRITE_SIZE (KB) over EVERY
dispatch of the mapping kernels (k_ascii_to_2bit, k_pe_*, bins, stats) and divides by the steps run
(warm-up + timed).  usage: pe_traffic.py gpurun_out/prof_<tag> <steps incl. warm-up>"""
import csv
import glob
import json
import os
import sys

out, steps = sys.argv[1], int(sys.argv[2])
want = ("k_pe_", "k_ascii_to_2bit", "k_bin_", "k_reduce_stats")
tot = {"TCC_EA0_RDREQ_sum": 0.0, "TCC_EA0_RDREQ_128B_sum": 0.0, "WRITE_SIZE": 0.0}
for f in sorted(glob.glob(os.path.join(out, "pmc*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in tot and any(w in r["Kernel_Name"] for w in want):
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
rd, wr = tot["TCC_EA0_RDREQ_sum"] / steps, tot["WRITE_SIZE"] / steps
res = {"hbm_bytes_per_step": rd * 128 + wr * 1024, "read_requests_128B": rd, "write_kb": wr,
       "share_128B": tot["TCC_EA0_RDREQ_128B_sum"] / max(tot["TCC_EA0_RDREQ_sum"], 1), "steps": steps, "source": out}
print(json.dumps(res))
