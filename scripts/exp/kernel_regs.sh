#!/bin/bash
# registers / LDS / spills of every kernel in an object file: scripts/exp/kernel_regs.sh attention [pattern]
D=$(mktemp -d); cp $(dirname $0)/../../bert_multimodal_transformer_amd/lib/obj/$1.o $D/x.o; cd $D
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading x.o > /dev/null 2>&1
f=$(ls | grep gfx950 | head -1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $f | awk '/ \.name:/ {n=$2} /\.vgpr_count:/ {v=$2} /\.agpr_count:/ {a=$2} /\.vgpr_spill_count:/ {s=$2} /\.group_segment_fixed_size:/ {l=$2} /\.wavefront_size:/ {printf "%-110s vgpr %s agpr %s spill %s lds %s\n", substr(n,1,110), v, a, s, l}' | grep -E "${2:-.}"
rm -rf $D
