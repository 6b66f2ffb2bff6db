#!/bin/bash
# round-3 probe 2: isolated units kept as units (general path), larger slot windows + private row-blocks (symmetric path)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r03c; mkdir -p $OUT; cd $ROOT
S=$OUT/probe2.md
python tools/abl.py syn-nlpkkt --edge 120 --header default: > $S 2>$OUT/err.txt
python tools/abl.py syn-nlpkkt --edge 120 --symmetric w512:spx.gpu.sym_wide_rows=512 w1024:spx.gpu.sym_wide_rows=1024 w1536:spx.gpu.sym_wide_rows=1536 w2048:spx.gpu.sym_wide_rows=2048 w2048x4:spx.gpu.sym_wide_rows=2048,spx.gpu.waves=4 >> $S 2>>$OUT/err.txt
python tools/abl.py syn-nlpkkt --edge 240 --steps 30 default: >> $S 2>>$OUT/err.txt
python tools/abl.py syn-nlpkkt --edge 240 --steps 30 --symmetric w1024:spx.gpu.sym_wide_rows=1024 w2048:spx.gpu.sym_wide_rows=2048 >> $S 2>>$OUT/err.txt
python tools/abl.py syn-kkt2f --edge 120 --symmetric default: w2048:spx.gpu.sym_wide_rows=2048 >> $S 2>>$OUT/err.txt
python tools/abl.py syn-nd24k --symmetric default: >> $S 2>>$OUT/err.txt
python tools/abl.py syn-cant default: >> $S 2>>$OUT/err.txt
python tools/abl.py syn-cant --symmetric default: >> $S 2>>$OUT/err.txt
cat $S; tail -n 3 $OUT/err.txt
(time python -m pytest tests -m gpu -x -q -n 4) > $OUT/pytest.log 2>&1; tail -n 5 $OUT/pytest.log
