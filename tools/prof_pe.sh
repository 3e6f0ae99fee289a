#!/bin/bash
# Per-kernel totals of one paired-end bench run (GPU box): bash tools/prof_pe.sh <tag> [env ...] -- [bench args]
set -u
TAG=$1; shift
while [ $# -gt 0 ] && [ "$1" != "--" ]; do export "$1"; shift; done
[ $# -gt 0 ] && shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
OUT=gpurun_out/pe_$TAG; mkdir -p $OUT
ARGS="--mode pe --no-extra --no-cpu-baseline --steps 2 --warmup 1 $*"
rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/trace -o t -- python3 bench.py $ARGS > $OUT/trace.json 2> $OUT/trace.log || { tail -5 $OUT/trace.log; exit 1; }
python3 tools/trace_sum.py $OUT/trace 3 | tee $OUT/trace_sum.txt
python3 - $OUT/trace <<'PY' | tee $OUT/timeline.txt
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "walt::" in r["Kernel_Name"] and any(x in r["Kernel_Name"] for x in ("k_pe_", "k_ascii", "k_bin", "k_reduce"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# wall span of the last step and the union of busy time
last = [i for i, r in enumerate(rows) if "k_ascii_to_2bit" in r["Kernel_Name"]]
# a step makes several passes, each with two ascii kernels; take the final third of the launches
n = len(rows)
sel = rows[2 * n // 3:]
t0, t1 = int(sel[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in sel)
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel)
busy = 0; cur_s, cur_e = ev[0]
for s, e in ev[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e in ev)
print("last step: span %.1f ms, union of kernel time %.1f ms, sum of kernel time %.1f ms, %d launches" % ((t1 - t0) / 1e6, busy / 1e6, tot / 1e6, len(sel)))
hist = collections.Counter()
for r in sel:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    nm = r["Kernel_Name"].split("(")[0].replace("void walt::", "")
    hist[(nm, "<0.05" if d < 0.05 else "<0.3" if d < 0.3 else "<1" if d < 1 else "<3" if d < 3 else ">=3")] += 1
for k in sorted(hist): print("  %-40s %-6s %d" % (k[0][:40], k[1], hist[k]))
PY
rm -rf $OUT/trace
