#!/bin/bash
# round-3 probe 19: a y tile per wavefront on the bench matrix (spx.gpu.wave_tiles), in-process A/B
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r03t; mkdir -p $OUT; cd $ROOT
S=$OUT/probe19.md
python tools/abl.py syn-nlpkkt --edge 240 --steps 30 --header shared:spx.gpu.wave_tiles=false,spx.gpu.waves=4 tiles:spx.gpu.wave_tiles=true,spx.gpu.waves=4 shared2:spx.gpu.wave_tiles=false,spx.gpu.waves=4 tiles2:spx.gpu.wave_tiles=true,spx.gpu.waves=4 tiles8:spx.gpu.wave_tiles=true,spx.gpu.waves=8 auto: > $S 2>$OUT/err.txt
python tools/abl.py syn-nlpkkt --edge 120 shared:spx.gpu.wave_tiles=false,spx.gpu.waves=4 tiles:spx.gpu.wave_tiles=true,spx.gpu.waves=4 shared2:spx.gpu.wave_tiles=false,spx.gpu.waves=4 tiles2:spx.gpu.wave_tiles=true,spx.gpu.waves=4 >> $S 2>>$OUT/err.txt
cat $S; tail -n 2 $OUT/err.txt
