#!/bin/bash
# probe for a two-way split-K of the N = 768, K >= 2304 launches: what does ONE block of it run (half the k range, 128 x 128 tile, one block per CU)?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for nset in 6 24; do
echo "== full k range, default selection (64 x 64 tiles), $nset operand sets"; timeout 60 tools/bin/gemm_bench --nset $nset --only "768" 2>&1 | grep -E "ffn2 |ffn1 |dgrad qkv"
echo "== half k range, 128 x 128 tiles (114 blocks)"; timeout 60 tools/bin/gemm_bench --nset $nset --only probe --tile 128 2>&1 | grep probe
echo "== half k range, 64 x 64 tiles (456 blocks)"; timeout 60 tools/bin/gemm_bench --nset $nset --only probe 2>&1 | grep probe
echo "== full k range, 128 x 128 tiles (114 blocks)"; timeout 60 tools/bin/gemm_bench --nset $nset --tile 128 2>&1 | grep -E "ffn2 |ffn1 |dgrad qkv"
done
