#!/bin/bash
# Diagnostic builds of libmoe_gfx950 with ablation / tuning macros (cdna guide rule 17/24: A/B in one process).
# usage: tools/build_variants.sh name1:-DFLAG1 name2:"-DFLAG2 -DFLAG3" ...
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/variants
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  [ "$flags" = "$spec" ] && flags=""
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Wno-unused-function $flags \
     moe_llava_amd/csrc/moe_routing.hip moe_llava_amd/csrc/moe_routing_top1.hip moe_llava_amd/csrc/moe_gemm.hip moe_llava_amd/csrc/moe_gemv.hip moe_llava_amd/csrc/moe_wgrad.hip moe_llava_amd/csrc/moe_backward.hip moe_llava_amd/csrc/moe_block.hip -o tools/variants/libmoe_$name.so &
done
wait
ls -la tools/variants
