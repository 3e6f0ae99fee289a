#!/bin/bash
# Diagnostic library on a GPU box: what parts of k_se_stage cost -- the run's gain when a part runs twice.
#   bash tools/twice.sh <tag> [bench args...]   (bits: 1 fence search, 2 position fetch, 4 candidate list, 8 entry loads)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
OUT=gpurun_out/twice_$TAG; mkdir -p $OUT
export WALT_AMD_LIB=$R/walt_amd/lib/libwalt_amd_diag.so
for T in 0 1 2 4 8; do
  WALT_AMD_TWICE=$T timeout -k 10 400 python3 bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline --opt se_pipe=0 "$@" > $OUT/t$T.json 2> $OUT/t$T.log || { tail -3 $OUT/t$T.log; exit 1; }
  python3 - $OUT/t$T.json $T <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["kernel_ms"]
print("twice=%s map %.2f groups %s" % (sys.argv[2], k["map_se"], {a: round(b, 2) for a, b in k["by_group"].items()}))
PY
done
WALT_AMD_TWICE=65536 WALT_AMD_STAMPS=4 timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline --opt se_pipe=0 "$@" 2>&1 >/dev/null | grep "k_se_stage"
