#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for AB in 4 0; do for G in 0 1280 2560 5120; do
  WALT_AMD_ABLATE=$AB WALT_AMD_GRID=$G timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | grep -E "^\{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate=$AB grid=$G', d['kernel_ms'])"
done; done
