#!/bin/bash
# round-3 probe 18: joined row-blocks of 16 k / 32 k nonzeros on the general path (spx.gpu.rowblock_elems beyond 8192), in-process A/B
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r03s; mkdir -p $OUT; cd $ROOT
S=$OUT/probe18.md
python tools/abl.py syn-nlpkkt --edge 120 --header default: j16k:spx.gpu.rowblock_elems=16384,spx.gpu.rowblock_rows=2048 j32k:spx.gpu.rowblock_elems=32768,spx.gpu.rowblock_rows=2048 j16kw8:spx.gpu.rowblock_elems=16384,spx.gpu.rowblock_rows=2048,spx.gpu.waves=8 j32kw8:spx.gpu.rowblock_elems=32768,spx.gpu.rowblock_rows=2048,spx.gpu.waves=8 default2: > $S 2>$OUT/err.txt
python tools/abl.py syn-nlpkkt --edge 240 --steps 30 default: j16k:spx.gpu.rowblock_elems=16384,spx.gpu.rowblock_rows=2048 j32kw8:spx.gpu.rowblock_elems=32768,spx.gpu.rowblock_rows=2048,spx.gpu.waves=8 >> $S 2>>$OUT/err.txt
cat $S; tail -n 2 $OUT/err.txt
