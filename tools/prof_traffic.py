#!/usr/bin/env python3
"""Summarise one leg of tools/prof_legs.sh: per-kernel average duration (rocprofv3 --stats) and HBM traffic per
step from the PMC passes, written as <dir>/summary.md, <dir>/kernel_stats.csv and <dir>/traffic_entry.json;
tools/prof_collect.py files them under profiles/ (profiles/traffic.json[<leg>] is what bench.py's roofline.traffic
reads).

Traffic per step = sum over the walt:: kernels of one step of
    TCC_EA0_RDREQ_128B x 128 + TCC_EA0_RDREQ_64B x 64 + (RDREQ - 128B - 64B) x 32   bytes read
  + WRITE_SIZE x 1024                                                              bytes written
(MI355X_MICROARCH.md "HBM": FETCH_SIZE tallies 128-byte requests at 64 bytes on gfx950, so the request counters are
used by size; these are the L2's memory-side requests, Infinity-Cache hits included).  PMC values are summed over
every dispatch of the run and divided by the number of steps the run made (warmup + timed)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

out, leg, tag = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bench = json.loads(open(os.path.join(out, "bench.json")).read().strip().splitlines()[-1])
steps = bench["steps"] + bench["warmup"]
lines = ["# rocprofv3 summary, leg %s (%s)\n" % (leg, tag), "command: bench.py %s  (%d steps incl. warmup)\n" % (
    bench["config"]["workload"], steps), "bench line: value %.4g %s, %.3f ms/step\n" % (bench["value"], bench["unit"], bench["ms_per_step"])]


def is_ours(name):
    return "walt::" in name


st = glob.glob(os.path.join(out, "stats", "*kernel_stats.csv"))
kern_ms = {}
if st:
    lines.append("\n| kernel | calls | avg ms | min ms | max ms | total ms / step |\n|---|---|---|---|---|---|")
    for r in csv.DictReader(open(st[0])):
        if is_ours(r["Name"]):
            nm = r["Name"].split("(")[0].replace("void ", "")
            kern_ms[nm] = float(r["TotalDurationNs"]) / 1e6 / steps
            if float(r["TotalDurationNs"]) / 1e6 / steps < 0.05 and "k_map" not in nm and "k_pe" not in nm and "k_se_" not in nm:
                continue
            lines.append("| %s | %s | %.3f | %.3f | %.3f | %.3f |" % (nm, r["Calls"], float(r["AverageNs"]) / 1e6,
                                                                   float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6,
                                                                   float(r["TotalDurationNs"]) / 1e6 / steps))
    shutil.copy(st[0], os.path.join(out, "kernel_stats.csv"))


def pmc(sub):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(out, sub, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not is_ours(k):
                continue
            # index-building kernels run once, before the steps: not part of a step's traffic
            if not any(x in k for x in ("k_map_se", "k_se_stage", "k_se_verify", "k_se_tail", "k_lit_", "k_pe_", "k_ascii_to_2bit", "k_bin_", "k_reduce_stats")):
                continue
            tot[k.split("(")[0].replace("void ", "")][r["Counter_Name"]] += float(r["Counter_Value"])
    return tot


rd, wr, tlb = pmc("pmc_rd"), pmc("pmc_wr"), pmc("pmc_tlb")
read_b = write_b = 0.0
lines.append("\n| kernel | read requests / step | of which 128 B | read GB / step | written GB / step | UTCL1 miss / request |\n|---|---|---|---|---|---|")
for nm in sorted(set(rd) | set(wr)):
    d = rd.get(nm, {})
    q, q128, q64 = d.get("TCC_EA0_RDREQ_sum", 0.0), d.get("TCC_EA0_RDREQ_128B_sum", 0.0), d.get("TCC_EA0_RDREQ_64B_sum", 0.0)
    rb = (q128 * 128 + q64 * 64 + max(0.0, q - q128 - q64) * 32) / steps
    wb = wr.get(nm, {}).get("WRITE_SIZE", 0.0) * 1024 / steps
    t = tlb.get(nm, {})
    miss = t.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0.0) / t["TCP_UTCL1_REQUEST_sum"] if t.get("TCP_UTCL1_REQUEST_sum") else float("nan")
    read_b += rb
    write_b += wb
    lines.append("| %s | %.4g | %.4g | %.2f | %.2f | %.2f |" % (nm, q / steps, q128 / steps, rb / 1e9, wb / 1e9, miss))
total = read_b + write_b
step_s = bench["ms_per_step"] / 1e3
kernel_s = None
if leg.startswith("se") and bench.get("kernel_ms"):
    kernel_s = bench["kernel_ms"]["map_se"] / 1e3
lines.append("\nHBM traffic per step: %.2f GB read + %.2f GB written = **%.2f GB**" % (read_b / 1e9, write_b / 1e9, total / 1e9))
if kernel_s:
    lines.append("mapping kernels (HIP events of the profiled run): %.3f ms -> %.2f TB/s = %.3f of the 8 TB/s peak" % (
        kernel_s * 1e3, total / kernel_s / 1e12, total / kernel_s / 8e12))
else:
    lines.append("whole step: %.3f ms -> %.2f TB/s = %.3f of the 8 TB/s peak" % (step_s * 1e3, total / step_s / 1e12, total / step_s / 8e12))
text = "\n".join(lines) + "\n"
open(os.path.join(out, "summary.md"), "w").write(text)
print(text)

cfg = bench["config"]
sys.path.insert(0, root)
import bench as _bench  # csrc_sha(): the counters belong to these kernel sources (bench.py drops roofline.traffic when they change)
entry = {"genome": cfg.get("genome"), "genome_bp": cfg.get("genome_bp"), "source": "profiles/%s_%s_summary.md" % (tag, leg),
         "csrc_sha": _bench.csrc_sha(),
         "read_bytes": read_b, "write_bytes": write_b, "ms_per_step_under_profiler": bench["ms_per_step"]}
if leg.startswith("se"):
    entry.update({"hbm_bytes_per_launch": total, "reads_per_launch": cfg.get("reads_per_gpu")})
else:
    entry.update({"hbm_bytes_per_step": total, "pairs_per_step": cfg.get("pairs_per_gpu")})
json.dump(entry, open(os.path.join(out, "traffic_entry.json"), "w"), indent=1, sort_keys=True)
# tools/prof_collect.py (build container, after gpurun merged gpurun_out/) copies summary.md / kernel_stats.csv into
# profiles/ and merges the entry into profiles/traffic.json
