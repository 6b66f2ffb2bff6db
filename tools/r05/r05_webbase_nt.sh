#!/bin/bash
# round-5: syn-webbase, the one combination round 4 left unmeasured -- non-temporal loads of the matrix stream
# (values, column offsets, rows of the pieces) together with column slices on XCD groups: time (one process per
# library, tools/abl.py) and FETCH_SIZE / WRITE_SIZE per variant (rocprofv3 --pmc, one pass each)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05w; mkdir -p $OUT; cd $ROOT
NT=$ROOT/sparsex_amd/lib/variants/libsparsex_NT_STREAM.so
SETS="plain:spx.gpu.col_phases=1 c2:spx.gpu.col_phases=c2 c4:spx.gpu.col_phases=c4 auto:"
R=$OUT/webbase_nt.md; : > $R
echo "### regular loads" >> $R
timeout 600 python3 tools/abl.py syn-webbase --steps 400 --header $SETS 2>$OUT/abl_plain.err | tee -a $R
echo "### non-temporal stream loads (experiment build -DSPX_EXPERIMENT_NT_STREAM)" >> $R
SPX_LIB_PATH=$NT timeout 600 python3 tools/abl.py syn-webbase --steps 400 $SETS 2>$OUT/abl_nt.err | tee -a $R
for lib in plain nt; do
  for ph in 1 c2 c4; do
    for ctr in FETCH_SIZE WRITE_SIZE; do
      if [ $lib = nt ]; then export SPX_LIB_PATH=$NT; else unset SPX_LIB_PATH; fi
      bash tools/pmc.sh r05w/pmc_${lib}_${ph}_${ctr} "$ctr" --workload syn-webbase --opt spx.gpu.col_phases=$ph > /dev/null 2>&1
      echo "$lib col_phases=$ph $ctr: $(grep -h 'csx_spmv' $OUT/pmc_${lib}_${ph}_${ctr}/pmc_summary.txt | tr -s ' ' | tr '\n' ';')" | tee -a $OUT/pmc.txt
    done
  done
done
unset SPX_LIB_PATH
