#!/bin/bash
# round 6, VERDICT r05 item 3(b): joined row-blocks as the launch tuner's choice on the bench matrix (general path):
# tuner default against the joined trial switched off by pinning the row-block size, twice each
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06j; mkdir -p $OUT; cd $ROOT
R=$OUT/joined_tuner.md; : > $R
timeout 1500 python3 tools/abl.py syn-nlpkkt --edge 240 --steps 50 --header default: plain8k:spx.gpu.rowblock_elems=8192 default2: plain8k2:spx.gpu.rowblock_elems=8192 2>$OUT/a.err | tee -a $R
timeout 600 python3 tools/abl.py syn-nlpkkt --edge 120 --steps 200 default: 2>>$OUT/a.err | tee -a $R
