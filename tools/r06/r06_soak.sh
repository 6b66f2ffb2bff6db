#!/bin/bash
# round-6 soak: the larger randomised matrices (GPU against CSR, with set entries / save / restore), the sliced and
# rectangular ones and the small ones, with the round's options (read-once pipeline, passes of their own) in the draws
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06k; mkdir -p $OUT; cd $ROOT
timeout 1800 python3 tools/soak_large.py ${L0:-0} ${L1:-160} --roundtrip 2>&1 | tail -8 > $OUT/soak_large.txt
timeout 900 python3 tools/soak_slices.py ${S0:-0} ${S1:-60} 2>&1 | tail -5 > $OUT/soak_slices.txt
timeout 900 python3 tools/soak_rect.py ${R0:-0} ${R1:-80} 2>&1 | tail -5 > $OUT/soak_rect.txt
timeout 1200 python3 tools/soak_random.py ${Q0:-3000} ${Q1:-3800} 2>&1 | tail -5 > $OUT/soak_random.txt
for f in $OUT/*.txt; do echo "$(basename $f): $(tail -1 $f)"; done
