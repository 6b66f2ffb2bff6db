#!/bin/bash
# round-4 probe 12: leftover passes of a row-block bucket by bucket of the columns (spx.gpu.gather_sweep), A/B on
# syn-webbase in one process; then the GPU suite on the tree with the CSR partition fast path
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r04l; mkdir -p $OUT; cd $ROOT
R=$OUT/gather_sweep_raw.md; : > $R
python3 tools/abl.py syn-webbase --steps 400 --header "two column slices (round 3):spx.gpu.gather_sweep=0" "plain stream:spx.gpu.gather_sweep=0,spx.gpu.col_phases=1" \
   "plain + 4 buckets:spx.gpu.gather_sweep=4,spx.gpu.col_phases=1" "plain + 8 buckets:spx.gpu.gather_sweep=8,spx.gpu.col_phases=1" \
   "plain + 16 buckets:spx.gpu.gather_sweep=16,spx.gpu.col_phases=1" "plain + 32 buckets:spx.gpu.gather_sweep=32,spx.gpu.col_phases=1" \
   "c2 + 8 buckets:spx.gpu.gather_sweep=8,spx.gpu.col_phases=c2" "auto:" "two column slices (round 3):spx.gpu.gather_sweep=0" 2>/dev/null | tee -a $R
python3 tools/abl.py syn-bandrandom "sweep 0:spx.gpu.gather_sweep=0" "sweep 16:spx.gpu.gather_sweep=16" 2>/dev/null | tee -a $R
timeout 1800 python3 -X faulthandler -m pytest tests -x -q -m gpu -k "not multirank" -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
tail -4 $OUT/pytest_gpu.txt | cut -c1-200
