#!/bin/bash
# Paired-end kernels one at a time (WALT_AMD_PE_SERIAL=1): per-kernel time, HBM bytes read and written per step.
set -u
TAG=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
OUT=gpurun_out/pes_$TAG; mkdir -p $OUT
ARGS="--mode pe --no-extra --no-cpu-baseline --steps 2 --warmup 1 --opt pe_serial=1 $*"
rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/trace -o t -- python3 bench.py $ARGS > $OUT/trace.json 2> $OUT/trace.log || { tail -5 $OUT/trace.log; exit 1; }
python3 tools/trace_sum.py $OUT/trace 3 > $OUT/trace_sum.txt
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $R/$OUT/pmc_rd -o p -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_rd.log || { tail -5 $OUT/pmc_rd.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $R/$OUT/pmc_wr -o p -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_wr.log || { tail -5 $OUT/pmc_wr.log; exit 1; }
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
ms = {}
for ln in open(out + "/trace_sum.txt"):
    p = ln.split(None, 4)
    ms[p[4].strip()] = (float(p[0]), float(p[2]))
def pmc(sub):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(out + "/" + sub + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "walt::" not in k: continue
            nm = k.split("(")[0].replace("void ", "").replace("walt::", "")
            tot[nm][r["Counter_Name"]] += float(r["Counter_Value"])
    return tot
rd, wr = pmc("pmc_rd"), pmc("pmc_wr")
print("%-28s %9s %8s %9s %9s %8s" % ("kernel", "ms/step", "launch", "read GB", "write GB", "TB/s"))
for nm in sorted(ms, key=lambda k: -ms[k][0]):
    if not any(x in nm for x in ("k_pe_", "k_ascii")): continue
    r = rd.get(nm, {}).get("TCC_EA0_RDREQ_128B_sum", 0) * 128 / 3 / 1e9
    w = wr.get(nm, {}).get("WRITE_SIZE", 0) * 1024 / 3 / 1e9
    t = ms[nm][0]
    print("%-28s %9.2f %8.1f %9.1f %9.1f %8.2f" % (nm[:28], t, ms[nm][1], r, w, (r + w) / t if t else 0))
PY
rm -rf $OUT/trace $OUT/pmc_rd $OUT/pmc_wr
