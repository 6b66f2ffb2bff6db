#!/bin/bash
# round-5 probe 8: depth of the unit-pass pipeline (rounds in flight per wavefront) x wavefronts per workgroup
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05h; mkdir -p $OUT; cd $ROOT
timeout 1200 python3 -m pytest tests/test_gpu_unit_windows.py -q -x 2>&1 | tail -3 | tee $OUT/pytest.txt
R=$OUT/abl.md; : > $R
X="spx.gpu.unit_windows=true,spx.gpu.unit_window_doubles=3072"
B="spx.gpu.unit_windows=true,spx.gpu.rowblock_rows=2048,spx.gpu.rowblock_elems=32768,spx.gpu.unit_window_doubles=12000"
M="spx.gpu.unit_windows=true,spx.gpu.rowblock_rows=1024,spx.gpu.rowblock_elems=16384,spx.gpu.unit_window_doubles=8192"
SETS="off-w4:spx.gpu.unit_windows=false,spx.gpu.waves=4"
for d in 2 3 4; do for w in 2 4; do SETS="$SETS w$w-d$d:$X,spx.gpu.waves=$w,spx.gpu.unit_window_depth=$d"; done; done
for d in 2 3 4; do for w in 4 8; do SETS="$SETS 16k-w$w-d$d:$M,spx.gpu.waves=$w,spx.gpu.unit_window_depth=$d"; done; done
for d in 3 4; do SETS="$SETS 32k-w8-d$d:$B,spx.gpu.waves=8,spx.gpu.unit_window_depth=$d"; done
timeout 1800 python3 tools/abl.py syn-nlpkkt --edge 240 --steps 40 --header $SETS 2>$OUT/abl240.err | tee -a $R
