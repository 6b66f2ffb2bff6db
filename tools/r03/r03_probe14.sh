#!/bin/bash
# round-3 probe 14: inline descriptors on / off inside one process (spx.gpu.inline_desc)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r03o; mkdir -p $OUT; cd $ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_stream_layout.py -x -q -n 4 -m gpu > $OUT/pytest_parity.log 2>&1; tail -n 3 $OUT/pytest_parity.log
S=$OUT/probe14.md
python tools/abl.py syn-nlpkkt --edge 120 --header inline: load:spx.gpu.inline_desc=false inline2: load2:spx.gpu.inline_desc=false > $S 2>$OUT/err.txt
python tools/abl.py syn-nlpkkt --edge 120 --symmetric inline: load:spx.gpu.inline_desc=false inline2: load2:spx.gpu.inline_desc=false >> $S 2>>$OUT/err.txt
python tools/abl.py syn-nlpkkt --edge 240 --steps 30 inline: load:spx.gpu.inline_desc=false inline2: load2:spx.gpu.inline_desc=false >> $S 2>>$OUT/err.txt
python tools/abl.py syn-nlpkkt --edge 240 --steps 30 --symmetric inline: load:spx.gpu.inline_desc=false >> $S 2>>$OUT/err.txt
python tools/abl.py syn-kkt2f --edge 120 inline: load:spx.gpu.inline_desc=false >> $S 2>>$OUT/err.txt
python tools/abl.py syn-cant --steps 300 inline: load:spx.gpu.inline_desc=false inline2: load2:spx.gpu.inline_desc=false >> $S 2>>$OUT/err.txt
python tools/abl.py syn-nd24k --steps 300 inline: load:spx.gpu.inline_desc=false >> $S 2>>$OUT/err.txt
python tools/abl.py syn-nd24k --steps 300 --symmetric inline: load:spx.gpu.inline_desc=false >> $S 2>>$OUT/err.txt
cat $S; tail -n 3 $OUT/err.txt
