#!/bin/bash
# round-5: eight wavefronts per SIMD (64 VGPRs, 20 B scratch) TOGETHER with row-blocks narrow enough for four
# workgroups per CU -- round 4 measured the two separately
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05s; mkdir -p $OUT; cd $ROOT
R=$OUT/sym_w8.md; : > $R
SETS="default: wide768:spx.gpu.sym_wide_rows=768 wide512:spx.gpu.sym_wide_rows=512"
echo "amdgpu_waves_per_eu(8, 8) on csx_spmv_symseg_notile_kernel" >> $R
SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_SYM_W8.so timeout 1200 python3 tools/abl.py syn-nlpkkt --edge 240 --symmetric --steps 50 --header $SETS 2>$OUT/w8.err | tee -a $R
echo "product build" >> $R
timeout 1200 python3 tools/abl.py syn-nlpkkt --edge 240 --symmetric --steps 50 default: 2>$OUT/w8_default.err | tee -a $R
