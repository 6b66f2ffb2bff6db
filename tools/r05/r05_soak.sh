#!/bin/bash
# round-5 soak: the larger randomised matrices (GPU against CSR, with set entries / save / restore), the sliced and
# rectangular ones, with the unit-window options drawn as well
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05k2; mkdir -p $OUT; cd $ROOT
timeout 1500 python3 tools/soak_large.py 0 120 --roundtrip 2>&1 | tail -8 > $OUT/soak_large.txt
timeout 900 python3 tools/soak_slices.py 0 60 2>&1 | tail -5 > $OUT/soak_slices.txt
timeout 900 python3 tools/soak_rect.py 0 80 2>&1 | tail -5 > $OUT/soak_rect.txt
timeout 900 python3 tools/soak_random.py 2000 2600 2>&1 | tail -5 > $OUT/soak_random.txt
for f in $OUT/*.txt; do tail -1 $f; done
