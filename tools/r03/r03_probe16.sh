#!/bin/bash
# round-3 probe 16: hardware counters of the general kernel on syn-nlpkkt e120, full build and x loads compiled out
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r03q; mkdir -p $OUT; cd $ROOT
( cd /tmp && TMPDIR=/tmp rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(SQ|TA|TCP|TCC|TD|GRBM|SPI)_[A-Za-z0-9_]+" | sort -u > $OUT/counters_avail.txt ); wc -l $OUT/counters_avail.txt
bash tools/build_variant.sh NOX "-DSPX_ABL_NOX" > /dev/null 2>&1
R=$OUT/pmc_general_e120.txt; : > $R
for v in FULL NOX; do
  if [ $v = FULL ]; then unset SPX_LIB_PATH SPX_BENCH_ABLATION; else export SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_$v.so SPX_BENCH_ABLATION=1; fi
  for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
    echo "== $v: $C" >> $R
    rm -f $OUT/tmp_$v/pmc_summary.txt
    bash tools/pmc.sh r03q/tmp_$v "$C" --edge 120 --steps 20 2>&1 | grep -E "csx_spmv_kernel<4>|rror" | head -8 >> $R
  done
done
unset SPX_LIB_PATH SPX_BENCH_ABLATION
cat $R
