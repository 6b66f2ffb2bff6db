#!/bin/bash
# round 6: the general kernel with the unit windows under the launch order by strips across the planes
# (spx.gpu.band_order; the window plan now follows the order): does x get fetched fewer than three times?
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06b; mkdir -p $OUT; cd $ROOT
R=$OUT/band_order_xw.md; : > $R
P="spx.gpu.unit_windows=true,spx.gpu.rowblock_elems=8192,spx.gpu.waves=4"
timeout 1500 python3 tools/abl.py syn-nlpkkt --edge 240 --steps 50 --header plain:$P band:$P,spx.gpu.band_order=true plain2:$P band2:$P,spx.gpu.band_order=true 2>$OUT/a.err | tee -a $R
