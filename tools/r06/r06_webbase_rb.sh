#!/bin/bash
# round 6: syn-webbase, row-block sizes around the launch tuner's choice: does a size whose workgroups fit the chip in
# ONE round (<= 1024 workgroups of eight wavefronts) beat the tuner's 1382?
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06wb; mkdir -p $OUT; cd $ROOT
R=$OUT/webbase_rowblocks.md; : > $R
SETS="default: e3000:spx.gpu.rowblock_elems=3000 e3600:spx.gpu.rowblock_elems=3600 e4096:spx.gpu.rowblock_elems=4096 e5000:spx.gpu.rowblock_elems=5000 e6000:spx.gpu.rowblock_elems=6000 e8192:spx.gpu.rowblock_elems=8192 e4096r1024:spx.gpu.rowblock_elems=4096,spx.gpu.rowblock_rows=1024 e6000r2048:spx.gpu.rowblock_elems=6000,spx.gpu.rowblock_rows=2048 default2:"
timeout 900 python3 tools/abl.py syn-webbase --steps 300 --header $SETS 2>$OUT/a.err | tee -a $R
