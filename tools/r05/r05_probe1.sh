#!/bin/bash
# round-5 probe 1: the general kernel with the unit windows of x in LDS and pipelined unit passes
# (csx_spmv_xw_kernel): its GPU parity tests, then A/B against the plain kernel inside one process
# (tools/abl.py) on the bench matrix at edge 120 and 240, wavefront counts and row-block sizes pinned.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05a; mkdir -p $OUT; cd $ROOT
timeout 900 python3 -m pytest tests/test_gpu_unit_windows.py -x -q 2>&1 | tail -15 > $OUT/pytest.txt
cat $OUT/pytest.txt
R=$OUT/abl.md; : > $R
SETS="off:spx.gpu.unit_windows=false auto: \
 off-w4:spx.gpu.unit_windows=false,spx.gpu.waves=4 \
 on-w4:spx.gpu.unit_windows=true,spx.gpu.waves=4 \
 on-w8:spx.gpu.unit_windows=true,spx.gpu.waves=8 \
 on-w4-16k:spx.gpu.unit_windows=true,spx.gpu.waves=4,spx.gpu.rowblock_elems=16384,spx.gpu.unit_window_doubles=8192 \
 on-w8-16k:spx.gpu.unit_windows=true,spx.gpu.waves=8,spx.gpu.rowblock_elems=16384,spx.gpu.unit_window_doubles=8192 \
 off-w4-16k:spx.gpu.unit_windows=false,spx.gpu.waves=4,spx.gpu.rowblock_elems=16384 \
 on-w4-4k:spx.gpu.unit_windows=true,spx.gpu.waves=4,spx.gpu.rowblock_elems=4096 \
 on-w2-4k:spx.gpu.unit_windows=true,spx.gpu.waves=2,spx.gpu.rowblock_elems=4096 \
 off-again:spx.gpu.unit_windows=false"
timeout 900 python3 tools/abl.py syn-nlpkkt --edge 120 --header $SETS 2>$OUT/abl120.err | tee -a $R
timeout 1500 python3 tools/abl.py syn-nlpkkt --edge 240 --steps 40 $SETS 2>$OUT/abl240.err | tee -a $R
tail -5 $OUT/abl120.err $OUT/abl240.err
