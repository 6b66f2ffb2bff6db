#!/bin/bash
# round 3: extended randomised parity with the round's options in the draw (column slices, kept units, band order)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r03_soak; mkdir -p $OUT; cd $ROOT
( timeout 900 python tools/soak_random.py 0 700 > $OUT/random_a.log 2>&1; tail -n 3 $OUT/random_a.log ) &
( timeout 900 python tools/soak_random.py 700 1400 > $OUT/random_b.log 2>&1; tail -n 3 $OUT/random_b.log ) &
( timeout 1200 python tools/soak_large.py 0 150 > $OUT/large_a.log 2>&1; tail -n 2 $OUT/large_a.log ) &
( timeout 1200 python tools/soak_large.py 150 300 > $OUT/large_b.log 2>&1; tail -n 2 $OUT/large_b.log ) &
( timeout 1200 python tools/soak_large.py 1000 1080 --roundtrip > $OUT/roundtrip.log 2>&1; tail -n 2 $OUT/roundtrip.log ) &
wait
grep -h "FAILED" $OUT/*.log | head -20
grep -ch " ok" $OUT/large_a.log $OUT/large_b.log $OUT/roundtrip.log
