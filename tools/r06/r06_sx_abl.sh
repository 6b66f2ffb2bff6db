#!/bin/bash
# round 6: what the pipelined read-once kernel spends its time on (e240, symmetric path, spx.gpu.sym_pipeline=true,
# 8 wavefronts): experiment builds that leave one thing out each (results wrong on purpose, the rows say so)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06sxa; mkdir -p $OUT; cd $ROOT
R=$OUT/sx_ablation.md; : > $R
export SPX_SX_PASSES_PER_ROUND=${SPX_SX_PASSES_PER_ROUND:-1}
SET="sx:spx.gpu.sym_pipeline=true,spx.gpu.waves=8"
run() { timeout 900 python3 tools/abl.py syn-nlpkkt --edge 240 --symmetric --steps 50 "$@"; }
run --header "$SET" 2>$OUT/abl_default.err | tee -a $R
for v in ${VARIANTS:-SYM_NOOWN SYM_NOHANDOVER SYM_NOINIT SYM_NOWRITES SYM_NOX SYM_NOSLOTADD SYM_STREAM SYM_STREAM_NOWRITES}; do
  echo "$v" >> $R
  SPX_BENCH_ABLATION=1 SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_$v.so run "$SET" 2>$OUT/abl_$v.err | tee -a $R
done
run "$SET" 2>>$OUT/abl_default.err | tee -a $R
