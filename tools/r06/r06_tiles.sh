#!/bin/bash
# round 6: tile passes with their descriptors fetched a pass ahead (symtile_run) against the former form
# (variant SYM_NOTILERUN), syn-nd24k symmetric (+ syn-cant symmetric, which holds tiles and segments);
# parity on the GPU tests that run tile kernels first
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06t; mkdir -p $OUT; cd $ROOT
timeout 1200 python3 -m pytest tests -m gpu -x -q -k "nd24k or tile or symmetric" > $OUT/pytest_tiles.txt 2>&1; tail -2 $OUT/pytest_tiles.txt
R=$OUT/tile_run_ab.md; : > $R
SETS="default: w4:spx.gpu.waves=4 w8:spx.gpu.waves=8 w2:spx.gpu.waves=2 lists:spx.gpu.sym_spill=lists atomic:spx.gpu.sym_spill=atomic"
for rep in 1 2; do
  echo "descriptors a pass ahead (product build)" >> $R
  timeout 600 python3 tools/abl.py syn-nd24k --symmetric --steps 300 --header $SETS 2>$OUT/a.err | tee -a $R
  echo "SYM_NOTILERUN (the former form)" >> $R
  SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_SYM_NOTILERUN.so timeout 600 python3 tools/abl.py syn-nd24k --symmetric --steps 300 $SETS 2>$OUT/b.err | tee -a $R
done
echo "syn-cant symmetric" >> $R
timeout 600 python3 tools/abl.py syn-cant --symmetric --steps 300 default: 2>>$OUT/a.err | tee -a $R
SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_SYM_NOTILERUN.so timeout 600 python3 tools/abl.py syn-cant --symmetric --steps 300 default: 2>>$OUT/b.err | tee -a $R
