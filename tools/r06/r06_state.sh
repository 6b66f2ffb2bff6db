#!/bin/bash
# round 6: where the four configurations stand with the tree as it is (tuner defaults), next to the plain read-once
# kernel on the bench matrix
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06s; mkdir -p $OUT; cd $ROOT
R=$OUT/state.md; : > $R
timeout 900 python3 tools/abl.py syn-nlpkkt --edge 240 --symmetric --steps 50 --header default: plain:spx.gpu.sym_pipeline=false default2: 2>$OUT/a.err | tee -a $R
timeout 600 python3 tools/abl.py syn-nd24k --symmetric --steps 300 default: default2: 2>>$OUT/a.err | tee -a $R
timeout 600 python3 tools/abl.py syn-cant --steps 300 default: 2>>$OUT/a.err | tee -a $R
timeout 600 python3 tools/abl.py syn-cant --symmetric --steps 300 default: 2>>$OUT/a.err | tee -a $R
timeout 600 python3 tools/abl.py syn-webbase --steps 300 default: 2>>$OUT/a.err | tee -a $R
timeout 900 python3 tools/abl.py syn-nlpkkt --edge 240 --steps 50 default: 2>>$OUT/a.err | tee -a $R
