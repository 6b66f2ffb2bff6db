#!/bin/bash
# round-5: what the read-once symmetric kernel spends its time on (e240, symmetric path): experiment builds of
# spmv_kernels.hip that leave one thing out each (results are wrong on purpose, the rows say so)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05s; mkdir -p $OUT; cd $ROOT
R=$OUT/sym_ablation.md; : > $R
timeout 900 python3 tools/abl.py syn-nlpkkt --edge 240 --symmetric --steps 50 --header default: 2>$OUT/abl_default.err | tee -a $R
for v in ${VARIANTS:-SYM_NOSLOTADD SYM_ONEADD SYM_NOX SYM_NOHANDOVER}; do
  echo "$v" >> $R
  SPX_BENCH_ABLATION=1 SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_$v.so timeout 900 python3 tools/abl.py syn-nlpkkt --edge 240 --symmetric --steps 50 default: 2>$OUT/abl_$v.err | tee -a $R
done
timeout 900 python3 tools/abl.py syn-nlpkkt --edge 240 --symmetric --steps 50 default: 2>>$OUT/abl_default.err | tee -a $R
