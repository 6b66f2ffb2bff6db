#!/bin/bash
# round-5: the matrix stream non-temporal in the unit-window kernel (values, descriptors): does x survive in the L2s
# from one plane of the grid to the next (TCC_MISS x 128 B = 7.6 GB per launch against 6.6 GB algorithmic)?
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05n; mkdir -p $OUT; cd $ROOT
NT=$ROOT/sparsex_amd/lib/variants/libsparsex_NT_ALL.so
R=$OUT/nt_xw.md; : > $R
SETS="off:spx.gpu.unit_windows=false on:spx.gpu.unit_windows=true"
timeout 900 python3 tools/abl.py syn-nlpkkt --edge 240 --steps 50 --header $SETS 2>$OUT/a.err | tee -a $R
echo "non-temporal stream (-DSPX_EXPERIMENT_NT_STREAM, both kernels)" >> $R
SPX_LIB_PATH=$NT timeout 900 python3 tools/abl.py syn-nlpkkt --edge 240 --steps 50 $SETS 2>$OUT/b.err | tee -a $R
timeout 900 python3 tools/abl.py syn-nlpkkt --edge 240 --steps 50 $SETS 2>>$OUT/a.err | tee -a $R
for lib in plain nt; do
  if [ $lib = nt ]; then export SPX_LIB_PATH=$NT; else unset SPX_LIB_PATH; fi
  bash tools/pmc.sh r05n/pmc_${lib} "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" --opt spx.gpu.unit_windows=true > /dev/null 2>&1
  echo "== $lib"; grep "xw_kernel<4>" $OUT/pmc_${lib}/pmc_summary.txt | cut -c1-120
done | tee $OUT/pmc.txt
unset SPX_LIB_PATH
