#!/bin/bash
# round 6: is syn-nd24k's symmetric step bound by a second, mostly empty round of workgroups (1934 row-blocks on
# 1792 workgroup slots at 7 per CU)?  The same generator at sizes around it: time per nonzero against row-blocks
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06n; mkdir -p $OUT; cd $ROOT
R=$OUT/nd24k_sizes.md; : > $R
H=--header
for sc in 0.70 0.80 0.88 0.92 0.96 1.00 1.05 1.15 1.30; do
  timeout 300 python3 tools/abl.py syn-nd24k --scale $sc --symmetric --steps 300 $H "s$sc:spx.gpu.waves=4,spx.gpu.sym_spill=atomic,spx.gpu.rowblock_elems=8192" 2>>$OUT/a.err | tee -a $R; H=
done
