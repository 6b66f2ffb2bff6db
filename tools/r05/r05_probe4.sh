#!/bin/bash
# round-5 probe 4: where a workgroup of the unit-window kernel spends its life (clock counters of the
# profile build), and SQ / TCP counters of the plain and the unit-window kernel on the bench matrix.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05d; mkdir -p $OUT; cd $ROOT
ON="on-w4:spx.gpu.unit_windows=true,spx.gpu.waves=4"
SPX_XW_PROFILE=1 SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_XW_PROFILE.so timeout 600 python3 tools/abl.py syn-nlpkkt --edge 240 --steps 40 --header "$ON" 2>$OUT/profile.err | tee $OUT/profile.md
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_DATA_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    for mode in false true; do
        bash tools/pmc.sh r05d/pmc_$mode "$set" --opt spx.gpu.unit_windows=$mode --opt spx.gpu.waves=4 > /dev/null 2>&1
    done
done
for mode in false true; do echo "== unit_windows=$mode"; cat $OUT/pmc_$mode/pmc_summary.txt; done | tee $OUT/pmc.txt
