#!/bin/bash
# round-3 probe 11: column slices in ONE launch, a group of XCDs per slice (spx.gpu.col_phases = c2 | c4 | c8)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r03l; mkdir -p $OUT; cd $ROOT
python -m pytest tests/test_gpu_parity.py -q -x -k "column_phases" > $OUT/pytest_phases.log 2>&1; tail -n 3 $OUT/pytest_phases.log
S=$OUT/probe11.md
python tools/abl.py syn-webbase --header --steps 300 plain:spx.gpu.col_phases=1 c2:spx.gpu.col_phases=c2 c4:spx.gpu.col_phases=c4 c8:spx.gpu.col_phases=c8 auto: c4w4:spx.gpu.col_phases=c4,spx.gpu.waves=4 c4w2:spx.gpu.col_phases=c4,spx.gpu.waves=2 c8w4:spx.gpu.col_phases=c8,spx.gpu.waves=4 c4e4k:spx.gpu.col_phases=c4,spx.gpu.rowblock_elems=4096 c4e1k:spx.gpu.col_phases=c4,spx.gpu.rowblock_elems=1024 c8e1k:spx.gpu.col_phases=c8,spx.gpu.rowblock_elems=1024 plain2:spx.gpu.col_phases=1 > $S 2>$OUT/err.txt
python tools/abl.py syn-bandrandom --steps 300 plain:spx.gpu.col_phases=1 c4:spx.gpu.col_phases=c4 >> $S 2>>$OUT/err.txt
python tools/abl.py syn-cant --steps 300 plain:spx.gpu.col_phases=1 c2:spx.gpu.col_phases=c2 >> $S 2>>$OUT/err.txt
cat $S; tail -n 3 $OUT/err.txt
