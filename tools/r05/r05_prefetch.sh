#!/bin/bash
# round-5: the unit-window kernel pulling the headers of a row-block its XCD starts D workgroups later into the L2
# (experiment builds -DSPX_EXPERIMENT_XW_PREFETCH=D): does the first round trip of a workgroup shorten?
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05f2; mkdir -p $OUT; cd $ROOT
R=$OUT/prefetch.md; : > $R
SETS="on:spx.gpu.unit_windows=true,spx.gpu.waves=4"
timeout 600 python3 tools/abl.py syn-nlpkkt --edge 240 --steps 50 --header $SETS 2>$OUT/a.err | tee -a $R
for D in ${DS:-16 48 160}; do
  echo "D = $D" >> $R
  SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_XW_PF$D.so timeout 600 python3 tools/abl.py syn-nlpkkt --edge 240 --steps 50 $SETS 2>$OUT/b$D.err | tee -a $R
done
timeout 600 python3 tools/abl.py syn-nlpkkt --edge 240 --steps 50 $SETS 2>>$OUT/a.err | tee -a $R
