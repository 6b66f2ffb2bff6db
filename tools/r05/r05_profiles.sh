#!/bin/bash
# round-5 evidence run, one box: the node's read roof in the SpMV's access pattern before and after, the whole
# profiles/<round>/ set (tools/refresh_profiles.sh), and counters of the bench kernel with and without the unit
# windows on THIS node (one rocprofv3 --pmc pass per counter group)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05p; mkdir -p $OUT; cd $ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/stream_pattern.hip -o $OUT/stream_pattern 2>/dev/null
timeout 300 $OUT/stream_pattern 64 30 2>&1 | tail -7 > $OUT/pattern_before.txt
bash tools/refresh_profiles.sh r05 > $OUT/refresh.log 2>&1
timeout 300 $OUT/stream_pattern 64 30 2>&1 | tail -7 > $OUT/pattern_mid.txt
for grp in "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "TCP_PENDING_STALL_CYCLES TCP_TA_DATA_STALL_CYCLES" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  bash tools/pmc.sh r05p/ctr_xw_$tag "$grp" > /dev/null 2>&1
  bash tools/pmc.sh r05p/ctr_plain_$tag "$grp" --opt spx.gpu.unit_windows=false > /dev/null 2>&1
done
bash tools/pmc.sh r05p/ctr_sym_SQ "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" --symmetric > /dev/null 2>&1
timeout 300 $OUT/stream_pattern 64 30 2>&1 | tail -7 > $OUT/pattern_after.txt
rm -f $OUT/stream_pattern
cat $OUT/pattern_before.txt $OUT/pattern_after.txt; tail -40 $OUT/refresh.log | cut -c1-220
