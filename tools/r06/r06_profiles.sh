#!/bin/bash
# round-6 evidence run, one box: the whole profiles/<round>/ set (tools/refresh_profiles.sh: rocprofv3 kernel stats,
# FETCH_SIZE / WRITE_SIZE passes and plain bench lines of the five configurations), and the counters of the
# symmetric step with and without the read-once pipeline on THIS node (one rocprofv3 --pmc pass per counter group)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06p; mkdir -p $OUT; cd $ROOT
bash tools/refresh_profiles.sh r06 > $OUT/refresh.log 2>&1
for grp in "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  bash tools/pmc.sh r06p/ctr_sx_$tag "$grp" --symmetric > /dev/null 2>&1
  bash tools/pmc.sh r06p/ctr_symplain_$tag "$grp" --symmetric --opt spx.gpu.sym_pipeline=false > /dev/null 2>&1
done
tail -40 $OUT/refresh.log | cut -c1-220
