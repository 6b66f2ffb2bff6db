#!/bin/bash
# round-5: (1) the general path's row-blocks joined side by side (longer-lived workgroups, more reuse of the staged
# x) under the unit-window kernel; (2) the symmetric kernel with everything but the stream left out
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05s; mkdir -p $OUT; cd $ROOT
R=$OUT/join.md; : > $R
SETS="default: j2:spx.gpu.rowblock_rows=1024,spx.gpu.rowblock_elems=16384 j2b6:spx.gpu.rowblock_rows=1024,spx.gpu.rowblock_elems=16384,spx.gpu.unit_window_doubles=6144 j4b6:spx.gpu.rowblock_rows=2048,spx.gpu.rowblock_elems=32768,spx.gpu.unit_window_doubles=6144 j4b8:spx.gpu.rowblock_rows=2048,spx.gpu.rowblock_elems=32768,spx.gpu.unit_window_doubles=8192 j3b6:spx.gpu.rowblock_rows=1024,spx.gpu.rowblock_elems=24576,spx.gpu.unit_window_doubles=6144"
timeout 1500 python3 tools/abl.py syn-nlpkkt --edge 240 --steps 50 --header $SETS 2>$OUT/join.err | tee -a $R
echo "symmetric: SPX_ABL_SYM_STREAM" >> $R
SPX_BENCH_ABLATION=1 SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_SYM_STREAM.so timeout 900 python3 tools/abl.py syn-nlpkkt --edge 240 --symmetric --steps 50 default: 2>$OUT/stream.err | tee -a $R
