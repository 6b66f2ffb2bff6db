#!/bin/bash
# round-3 probe 3: A/B of kept units (general), XCD split by values + no empty row-blocks (symmetric, wide row-blocks)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r03d; mkdir -p $OUT; cd $ROOT
S=$OUT/probe3.md
python tools/abl.py syn-nlpkkt --edge 120 --header keep: nokeep:spx.gpu.keep_units=false keep2: nokeep2:spx.gpu.keep_units=false > $S 2>$OUT/err.txt
python tools/abl.py syn-nlpkkt --edge 120 --symmetric w512:spx.gpu.sym_wide_rows=512 w1024:spx.gpu.sym_wide_rows=1024 w1536:spx.gpu.sym_wide_rows=1536 w2048:spx.gpu.sym_wide_rows=2048 nosegs:spx.gpu.sym_segments=false >> $S 2>>$OUT/err.txt
python tools/abl.py syn-nlpkkt --edge 240 --steps 30 keep: nokeep:spx.gpu.keep_units=false >> $S 2>>$OUT/err.txt
python tools/abl.py syn-nlpkkt --edge 240 --steps 30 --symmetric w1024:spx.gpu.sym_wide_rows=1024 w2048:spx.gpu.sym_wide_rows=2048 >> $S 2>>$OUT/err.txt
python tools/abl.py syn-kkt2f --edge 120 --symmetric default: w2048:spx.gpu.sym_wide_rows=2048 >> $S 2>>$OUT/err.txt
python tools/abl.py syn-kkt2f --edge 120 default: >> $S 2>>$OUT/err.txt
python tools/abl.py syn-nd24k --symmetric default: >> $S 2>>$OUT/err.txt
python tools/abl.py syn-cant keep: nokeep:spx.gpu.keep_units=false >> $S 2>>$OUT/err.txt
python tools/abl.py syn-webbase default: >> $S 2>>$OUT/err.txt
cat $S; tail -n 3 $OUT/err.txt
(time python -m pytest tests -m gpu -x -q -n 4) > $OUT/pytest.log 2>&1; tail -n 5 $OUT/pytest.log
