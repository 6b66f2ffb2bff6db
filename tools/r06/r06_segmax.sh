#!/bin/bash
# round 6: runs wider than four columns cut into segments the read-once pipeline can take (spx.gpu.sym_segment_max):
# syn-kkt2f (runs of six) and syn-cant on the symmetric path, tuner defaults otherwise
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06g; mkdir -p $OUT; cd $ROOT
R=$OUT/segment_max.md; : > $R
S="spx.gpu.sym_segments=true"
SETS="max8:$S max4:$S,spx.gpu.sym_segment_max=4 max3:$S,spx.gpu.sym_segment_max=3 max8again:$S max4again:$S,spx.gpu.sym_segment_max=4"
timeout 900 python3 tools/abl.py syn-kkt2f --edge 100 --symmetric --steps 200 --header $SETS 2>>$OUT/a.err | tee -a $R
timeout 900 python3 tools/abl.py syn-kkt2f --edge 160 --symmetric --steps 100 $SETS 2>>$OUT/a.err | tee -a $R
timeout 900 python3 tools/abl.py syn-nlpkkt --edge 140 --symmetric --steps 200 max8:$S max4:$S,spx.gpu.sym_segment_max=4 2>>$OUT/a.err | tee -a $R
