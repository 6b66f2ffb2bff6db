#!/bin/bash
# round 3: soak of the FINAL library (after inline descriptors, x pairs, column slices in one launch, joined row-blocks)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r03_soak3; mkdir -p $OUT; cd $ROOT
for k in 0 1 2 3; do ( timeout 1200 python tools/soak_random.py $((6200 + k * 1000)) $((7200 + k * 1000)) > $OUT/random_$k.log 2>&1; tail -n 1 $OUT/random_$k.log ) & done
for k in 0 1 2; do ( timeout 1200 python tools/soak_large.py $((1300 + k * 250)) $((1550 + k * 250)) > $OUT/large_$k.log 2>&1; tail -n 1 $OUT/large_$k.log ) & done
( timeout 1200 python tools/soak_large.py 2100 2300 --roundtrip > $OUT/roundtrip.log 2>&1; tail -n 1 $OUT/roundtrip.log ) &
( timeout 1200 python tools/soak_rect.py 1500 2700 > $OUT/rect.log 2>&1; tail -n 1 $OUT/rect.log ) &
wait
( timeout 1200 bash tools/soak_multirank.sh > $OUT/multirank.log 2>&1; tail -n 1 $OUT/multirank.log; grep -c "^ok" $OUT/multirank.log )
grep -h "FAILED\|FAIL " $OUT/*.log | head -20
