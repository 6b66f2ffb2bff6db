#!/bin/bash
# round 6: why is the strip launch order (spx.gpu.band_order) slower on the general path although it should keep x in
# the L2 between the planes?  FETCH_SIZE, L2 hits / misses and wait counters of the unit-window kernel, both orders
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06bp; mkdir -p $OUT; cd $ROOT
P="--opt spx.gpu.unit_windows=true --opt spx.gpu.rowblock_elems=8192 --opt spx.gpu.waves=4"
for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  tag=$(echo $grp | cut -d' ' -f1)
  bash tools/pmc.sh r06bp/plain_$tag "$grp" $P > /dev/null 2>&1
  bash tools/pmc.sh r06bp/band_$tag "$grp" $P --opt spx.gpu.band_order=true > /dev/null 2>&1
done
for f in $OUT/*/pmc_summary.txt; do echo "== $f"; grep xw_kernel $f; done
