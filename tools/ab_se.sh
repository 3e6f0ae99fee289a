#!/bin/bash
# A/B of single-end kernel variants on one GPU box: each variant is one bench.py run (own genome + index build) with
# its options; the JSON lines land in gpurun_out/ab_<tag>/<name>.json, stderr beside them.
#   bash tools/ab_se.sh <tag> "<name>|<env assignments>|<extra bench args>" ...
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
OUT=gpurun_out/ab_$TAG
mkdir -p $OUT
for SPEC in "$@"; do
  NAME=${SPEC%%|*}; REST=${SPEC#*|}; ENVS=${REST%%|*}; ARGS=${REST#*|}
  echo "== $NAME: env [$ENVS] args [$ARGS]"
  ( [ -n "$ENVS" ] && export $ENVS; timeout -k 10 900 python3 bench.py $ARGS > $OUT/$NAME.json 2> $OUT/$NAME.log ) || { echo "$NAME failed"; tail -n 5 $OUT/$NAME.log; exit 1; }
  python3 - "$OUT/$NAME.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d.get("kernel_ms", {})
print("   ms/step %.2f  value %.4g  map %.2f  groups %s  exact %s" % (
    d["ms_per_step"], d["value"], k.get("map_se", 0), {a: round(b, 2) for a, b in (k.get("by_group") or {}).items()},
    (d.get("cpu_baseline") or {}).get("bit_exact_vs_gpu")))
PY
done
