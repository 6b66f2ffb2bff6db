#!/bin/bash
# round 3: a second soak of the final library, fresh seed ranges (after probe 20; same kernels as r03_soak3)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r03_soak4; mkdir -p $OUT; cd $ROOT
for k in 0 1 2 3; do ( timeout 640 python tools/soak_random.py $((10200 + k * 1000)) $((11200 + k * 1000)) > $OUT/random_$k.log 2>&1; tail -n 1 $OUT/random_$k.log ) & done
for k in 0 1 2; do ( timeout 640 python tools/soak_large.py $((2300 + k * 250)) $((2550 + k * 250)) > $OUT/large_$k.log 2>&1; tail -n 1 $OUT/large_$k.log ) & done
( timeout 640 python tools/soak_large.py 3100 3300 --roundtrip > $OUT/roundtrip.log 2>&1; tail -n 1 $OUT/roundtrip.log ) &
( timeout 640 python tools/soak_rect.py 2700 3900 > $OUT/rect.log 2>&1; tail -n 1 $OUT/rect.log ) &
wait
( timeout 640 bash tools/soak_multirank.sh > $OUT/multirank.log 2>&1; tail -n 1 $OUT/multirank.log; grep -c "^ok" $OUT/multirank.log )
grep -h "FAILED\|FAIL " $OUT/*.log | head -20
