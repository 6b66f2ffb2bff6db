#!/bin/bash
# round-5: from the stripped unit-window kernel (no staging, no arithmetic) downwards -- what separates it from the
# pattern probe (6.4 TB/s on every node)?  Cumulative experiment builds; results wrong on purpose.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05b2; mkdir -p $OUT; cd $ROOT
R=$OUT/bisect.md; : > $R
SETS="${SETS:-on:spx.gpu.unit_windows=true,spx.gpu.waves=4}"
timeout 600 python3 tools/abl.py syn-nlpkkt --edge 240 --steps 50 --header $SETS 2>$OUT/a.err | tee -a $R
for v in ${VARIANTS:-BIS1 BIS2 BIS3 BIS4 BIS5}; do
  echo "$v" >> $R
  SPX_BENCH_ABLATION=1 SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_$v.so timeout 600 python3 tools/abl.py syn-nlpkkt --edge 240 --steps 50 $SETS 2>$OUT/$v.err | tee -a $R
done
timeout 600 python3 tools/abl.py syn-nlpkkt --edge 240 --steps 50 $SETS 2>>$OUT/a.err | tee -a $R
