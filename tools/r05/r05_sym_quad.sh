#!/bin/bash
# round-5: the read-once symmetric kernel with four passes in flight per wavefront (experiment build), and the
# row-block widths between the ones round 4 measured
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05s; mkdir -p $OUT; cd $ROOT
R=$OUT/sym_quad.md; : > $R
SETS="default: wide768:spx.gpu.sym_wide_rows=768 wide640:spx.gpu.sym_wide_rows=640 wide1280:spx.gpu.sym_wide_rows=1280"
timeout 1200 python3 tools/abl.py syn-nlpkkt --edge 240 --symmetric --steps 50 --header $SETS 2>$OUT/quad_default.err | tee -a $R
echo "four passes per round (-DSPX_EXPERIMENT_SYM_QUAD, 118 VGPRs)" >> $R
SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_SYM_QUAD.so timeout 1200 python3 tools/abl.py syn-nlpkkt --edge 240 --symmetric --steps 50 $SETS 2>$OUT/quad_quad.err | tee -a $R
SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_SYM_QUAD.so timeout 1200 python3 tools/abl.py syn-nd24k --symmetric --steps 200 default: 2>>$OUT/quad_quad.err | tee -a $R
timeout 1200 python3 tools/abl.py syn-nd24k --symmetric --steps 200 default: 2>>$OUT/quad_default.err | tee -a $R
