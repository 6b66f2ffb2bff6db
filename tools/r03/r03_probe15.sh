#!/bin/bash
# round-3 probe 15: four narrow unit passes side by side (spx.gpu.quad), in-process A/B; what auto picks
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r03p; mkdir -p $OUT; cd $ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -x -q -n 4 > $OUT/pytest_parity.log 2>&1; tail -n 3 $OUT/pytest_parity.log
S=$OUT/probe15.md
python tools/abl.py syn-nlpkkt --edge 120 --header pairs:spx.gpu.quad=false quad:spx.gpu.quad=true pairs2:spx.gpu.quad=false quad2:spx.gpu.quad=true auto: > $S 2>$OUT/err.txt
python tools/abl.py syn-nlpkkt --edge 240 --steps 30 pairs:spx.gpu.quad=false quad:spx.gpu.quad=true pairs2:spx.gpu.quad=false quad2:spx.gpu.quad=true >> $S 2>>$OUT/err.txt
python tools/abl.py syn-kkt2f --edge 120 pairs:spx.gpu.quad=false quad:spx.gpu.quad=true >> $S 2>>$OUT/err.txt
python tools/abl.py syn-cant --steps 300 pairs:spx.gpu.quad=false quad:spx.gpu.quad=true auto: >> $S 2>>$OUT/err.txt
python tools/abl.py syn-nd24k --steps 300 pairs:spx.gpu.quad=false quad:spx.gpu.quad=true >> $S 2>>$OUT/err.txt
python tools/abl.py syn-webbase --steps 300 pairs:spx.gpu.quad=false,spx.gpu.col_phases=1 quad:spx.gpu.quad=true,spx.gpu.col_phases=1 auto: >> $S 2>>$OUT/err.txt
cat $S; tail -n 3 $OUT/err.txt
