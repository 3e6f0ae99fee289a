#!/bin/bash
# A/B timing of two builds of libwalt_amd.so on ONE GPU box (boxes differ by a few per cent):
#   tools/ab.sh [bench args...]   -> alternates walt_amd/lib/libwalt_amd_A.so (A) and the current build (B)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
  for v in A B; do
    if [ $v = A ]; then export WALT_AMD_LIB=$R/walt_amd/lib/libwalt_amd_A.so; else unset WALT_AMD_LIB; fi
    python bench.py --no-cpu-baseline --steps 5 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), d.get('kernel_ms'))" || exit 1
  done
done
