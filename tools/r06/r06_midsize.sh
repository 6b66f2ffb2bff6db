#!/bin/bash
# round 6: does the tuner's order of decisions (wavefronts with the plain kernel first, then the read-once
# pipeline on / off) find the best pair on mid-size matrices?  Symmetric path, edges 100-180, kkt2f: tuner default
# against every forced pair
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06m; mkdir -p $OUT; cd $ROOT
R=$OUT/midsize_tuner.md; : > $R
S="spx.gpu.sym_segments=true"
SETS="auto:$S sx4:$S,spx.gpu.sym_pipeline=true,spx.gpu.waves=4 sx8:$S,spx.gpu.sym_pipeline=true,spx.gpu.waves=8 plain4:$S,spx.gpu.sym_pipeline=false,spx.gpu.waves=4 plain8:$S,spx.gpu.sym_pipeline=false,spx.gpu.waves=8 mirrored:spx.gpu.sym_segments=false default:"
H=--header
for e in 100 140 180; do
  timeout 900 python3 tools/abl.py syn-nlpkkt --edge $e --symmetric --steps 200 $H $SETS 2>>$OUT/a.err | tee -a $R; H=
done
timeout 900 python3 tools/abl.py syn-kkt2f --edge 100 --symmetric --steps 200 $SETS 2>>$OUT/a.err | tee -a $R
