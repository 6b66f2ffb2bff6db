#!/bin/bash
# L2 behaviour of the x gathers on syn-webbase: hit rate and fabric requests for a few column-panel widths
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_[A-Z0-9_]*\(RDREQ\|HIT\|MISS\|REQ\)[A-Za-z0-9_]*" | sort -u | head -60 > $ROOT/gpurun_out/tcc_counters.txt
cd $ROOT
for P in 0; do
  echo "== col_panel $P"
  bash tools/pmc.sh wb_p${P}_a "TCC_HIT_sum TCC_MISS_sum" --workload syn-webbase  2>&1 | grep csx_spmv
  bash tools/pmc.sh wb_p${P}_b "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" --workload syn-webbase  2>&1 | grep csx_spmv
  WORKLOADS=syn-webbase bash tools/quick_bench.sh 
done
