#!/bin/bash
# round-5 probe 5: the unit-window kernel against its LDS footprint (window budget) and the size of the
# row-blocks (planned row-blocks joined side by side), bench matrix, one process.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05e; mkdir -p $OUT; cd $ROOT
timeout 1200 python3 -m pytest tests/test_gpu_unit_windows.py -q -x 2>&1 | tail -5 | tee $OUT/pytest.txt
R=$OUT/abl.md; : > $R
X="spx.gpu.unit_windows=true"
SETS="off-w4:spx.gpu.unit_windows=false,spx.gpu.waves=4 \
 on-w4:$X,spx.gpu.waves=4 \
 on-w4-b3072:$X,spx.gpu.waves=4,spx.gpu.unit_window_doubles=3072 \
 on-w4-b2816:$X,spx.gpu.waves=4,spx.gpu.unit_window_doubles=2816 \
 on-w4-b2560:$X,spx.gpu.waves=4,spx.gpu.unit_window_doubles=2560 \
 on-w4-16k:$X,spx.gpu.waves=4,spx.gpu.rowblock_rows=1024,spx.gpu.rowblock_elems=16384,spx.gpu.unit_window_doubles=8192 \
 on-w8-16k:$X,spx.gpu.waves=8,spx.gpu.rowblock_rows=1024,spx.gpu.rowblock_elems=16384,spx.gpu.unit_window_doubles=8192 \
 on-w4-24k:$X,spx.gpu.waves=4,spx.gpu.rowblock_rows=1024,spx.gpu.rowblock_elems=24576,spx.gpu.unit_window_doubles=8192 \
 on-w8-24k:$X,spx.gpu.waves=8,spx.gpu.rowblock_rows=1024,spx.gpu.rowblock_elems=24576,spx.gpu.unit_window_doubles=8192 \
 on-w8-32k:$X,spx.gpu.waves=8,spx.gpu.rowblock_rows=2048,spx.gpu.rowblock_elems=32768,spx.gpu.unit_window_doubles=12000 \
 off-w4-16k:spx.gpu.unit_windows=false,spx.gpu.waves=4,spx.gpu.rowblock_rows=1024,spx.gpu.rowblock_elems=16384 \
 off-w8-16k:spx.gpu.unit_windows=false,spx.gpu.waves=8,spx.gpu.rowblock_rows=1024,spx.gpu.rowblock_elems=16384"
timeout 1500 python3 tools/abl.py syn-nlpkkt --edge 240 --steps 40 --header $SETS 2>$OUT/abl240.err | tee -a $R
