#!/bin/bash
# A/B of bench.py variants on one GPU box (any mode): bash tools/ab_any.sh <tag> "<name>|<env>|<bench args>" ...
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
OUT=gpurun_out/ab_$TAG
mkdir -p $OUT
for SPEC in "$@"; do
  NAME=${SPEC%%|*}; REST=${SPEC#*|}; ENVS=${REST%%|*}; ARGS=${REST#*|}
  echo "== $NAME: env [$ENVS] args [$ARGS]"
  ( export $ENVS; timeout -k 10 1100 python3 bench.py $ARGS > $OUT/$NAME.json 2> $OUT/$NAME.log ) || { echo "$NAME failed"; tail -n 8 $OUT/$NAME.log; exit 1; }
  python3 - "$OUT/$NAME.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
def show(x, ind=""):
    r = x.get("roofline") or {}
    c = x.get("cpu_baseline") or {}
    print("%s%-60s ms/step %8.2f  value %.4g  frac %s  exact %s" % (ind, x["metric"][:60], x.get("ms_per_step", 0), x.get("value", 0),
          ("%.3f" % r["frac"]) if r.get("frac") else None, c.get("bit_exact_vs_gpu")))
show(d)
for e in d.get("extra_lines", []):
    if "skipped" in e: print("   skipped:", e)
    else: show(e, "   ")
PY
done
