#!/bin/bash
# round 6: what the read-once passes that hold several units (9 % of the bench matrix' nonzeros; they run through the
# plain kernel's code inside csx_spmv_sx_kernel) cost: experiment build that skips them (INVALID results)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06x; mkdir -p $OUT; cd $ROOT
R=$OUT/sx_mixed_passes.md; : > $R
SET="sx:spx.gpu.sym_pipeline=true,spx.gpu.waves=8"
timeout 900 python3 tools/abl.py syn-nlpkkt --edge 240 --symmetric --steps 50 --header "$SET" 2>$OUT/a.err | tee -a $R
echo SYM_NOMIXED >> $R
SPX_BENCH_ABLATION=1 SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_SYM_NOMIXED.so timeout 900 python3 tools/abl.py syn-nlpkkt --edge 240 --symmetric --steps 50 "$SET" 2>>$OUT/a.err | tee -a $R
timeout 900 python3 tools/abl.py syn-nlpkkt --edge 240 --symmetric --steps 50 "$SET" 2>>$OUT/a.err | tee -a $R
