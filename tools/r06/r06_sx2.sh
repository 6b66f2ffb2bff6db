#!/bin/bash
# round 6: the pipelined read-once kernel against the traffic it causes beyond the algorithmic bytes (x fetched per
# plane of the grid, slot hand-overs): launch order by strips across the planes (spx.gpu.band_order), wider
# row-blocks (fewer hand-overs per row, fewer workgroups per CU), one / two passes per round
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06sx2; mkdir -p $OUT; cd $ROOT
R=$OUT/sx_order_and_width.md; : > $R
P="spx.gpu.sym_pipeline=true,spx.gpu.waves=8"
SETS="sx:$P band:$P,spx.gpu.band_order=true wide1536:$P,spx.gpu.sym_wide_rows=1536 wide2048:$P,spx.gpu.sym_wide_rows=2048 wide768:$P,spx.gpu.sym_wide_rows=768 band_wide2048:$P,spx.gpu.band_order=true,spx.gpu.sym_wide_rows=2048 plain_band:spx.gpu.sym_pipeline=false,spx.gpu.band_order=true sx_again:$P"
for b in 1 2; do
  echo "passes per round: $b" >> $R
  SPX_SX_PASSES_PER_ROUND=$b timeout 1500 python3 tools/abl.py syn-nlpkkt --edge 240 --symmetric --steps 50 --header $SETS 2>$OUT/abl_b$b.err | tee -a $R
done
