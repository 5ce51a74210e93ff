#!/bin/bash
# round 5, VERDICT r4 item 4a: what would fusing the out-projection's dgrad into the attention backward cost?  A COST PROBE, not an
# implementation: gpurun_ab/attnprobe = attention.hip built with -DMB_ATTN_FUSE_PROBE (scripts/build_variant.py attnprobe
# -DMB_ATTN_FUSE_PROBE --files=attention.hip): every (sample, head) block computes a 64 x 64 x 768 product for its dO image straight
# from global memory (both operands k-contiguous: the favourable case that assumes a transposed bf16 copy of Wo) instead of reading
# dCtx.  Against it: the stand-alone dgrad-out GEMM that the fusion would remove (tools/gemm_bench).  Same box, three times each.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
{
for rep in 1 2 3; do
  echo "== attn_bench, library as shipped"; timeout 60 tools/bin/attn_bench | grep "attention backward"
  echo "== attn_bench, attention backward computing its own dCtx tile (cost probe)"; LD_LIBRARY_PATH=$R/gpurun_ab/attnprobe:$LD_LIBRARY_PATH timeout 60 tools/bin/attn_bench | grep "attention backward"
done
echo "== the launch the fusion would remove (tools/gemm_bench, rotating operands)"; timeout 60 tools/bin/gemm_bench | grep "dgrad out"
echo "== C5 shape (B=32, L=128 uses the 8-wave kernel: the probe does not apply)"
} > $O/attn_fusion_probe.txt 2>&1
cat $O/attn_fusion_probe.txt
