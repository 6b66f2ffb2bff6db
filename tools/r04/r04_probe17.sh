#!/bin/bash
# round-4 probe 17: the read-once symmetric path at the contract size against the width of its row-blocks and the wavefront count
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r04q; mkdir -p $OUT; cd $ROOT
python3 tools/abl.py syn-nlpkkt --edge 240 --symmetric --header --steps 30 default: wide512:spx.gpu.sym_wide_rows=512 wide2048:spx.gpu.sym_wide_rows=2048 \
    waves4:spx.gpu.waves=4 wide512w4:spx.gpu.sym_wide_rows=512,spx.gpu.waves=4 wide256w4:spx.gpu.sym_wide_rows=256,spx.gpu.waves=4 default: 2>&1 | grep "^|" | tee $OUT/sym_wide_rows_raw.md
