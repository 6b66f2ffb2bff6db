#!/bin/bash
# round-5 probe 7: clock stamps of every workgroup of the unit-window kernel (profile build)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05g; mkdir -p $OUT; cd $ROOT
ON="on-w4:spx.gpu.unit_windows=true,spx.gpu.waves=4,spx.gpu.unit_window_doubles=3072"
ON8="on-w8-32k:spx.gpu.unit_windows=true,spx.gpu.waves=8,spx.gpu.rowblock_rows=2048,spx.gpu.rowblock_elems=32768,spx.gpu.unit_window_doubles=12000"
SPX_XW_PROFILE_OUT=/tmp/prof.bin SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_XW_PROFILE.so timeout 600 python3 tools/abl.py syn-nlpkkt --edge 240 --steps 40 --header "$ON" "$ON8" 2>$OUT/profile.err | tee $OUT/profile.md
tail -3 $OUT/profile.err
