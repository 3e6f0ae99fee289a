#!/bin/bash
# VGPR / SGPR-spill / LDS / scratch usage of every kernel of one .hip file (device-only compile, no GPU needed).
#   tools/kernel_resources.sh walt_amd/csrc/map_pe.hip [extra hipcc flags, e.g. -DWALT_ONLY_NW=7 -DWALT_SEEDPATTERN=5]
set -e
SRC=$(readlink -f "$1"); shift
T=$(mktemp -d)
cd "$(dirname "$SRC")"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -w "$@" -c "$SRC" -o $T/k.co
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/k.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.elf
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.elf | grep -E "\.name:|\.vgpr_count|spill_count|group_segment|private_segment" | paste - - - - - - | sed 's/  */ /g' | (c++filt 2>/dev/null || cat) | sed -E "s/\(walt::IndexView.*\)//" | sed -E 's/\.group_segment_fixed_size/lds/; s/\.private_segment_fixed_size/scratch/; s/\.sgpr_spill_count/sspill/; s/\.vgpr_count/vgpr/; s/\.vgpr_spill_count/vspill/; s/\.name: //' | cut -c1-260
rm -rf $T
