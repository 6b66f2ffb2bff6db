#!/bin/bash
# round 4: randomised parity soak of the final library on fresh seed ranges, the last tree (fused transform passes, values placed in parallel) (tools/soak_*.py: GPU against CSR)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r04_soak4; mkdir -p $OUT; cd $ROOT
for k in 0 1 2 3; do ( timeout 600 python tools/soak_random.py $((40000 + k * 1000)) $((41000 + k * 1000)) > $OUT/random_$k.log 2>&1; tail -n 1 $OUT/random_$k.log ) & done
for k in 0 1 2; do ( timeout 600 python tools/soak_large.py $((11000 + k * 250)) $((11250 + k * 250)) > $OUT/large_$k.log 2>&1; tail -n 1 $OUT/large_$k.log ) & done
( timeout 600 python tools/soak_large.py 12000 12200 --roundtrip > $OUT/roundtrip.log 2>&1; tail -n 1 $OUT/roundtrip.log ) &
( timeout 600 python tools/soak_rect.py 13000 14200 > $OUT/rect.log 2>&1; tail -n 1 $OUT/rect.log ) &
wait
( timeout 900 bash tools/soak_multirank.sh > $OUT/multirank.log 2>&1; tail -n 1 $OUT/multirank.log; grep -c "^ok" $OUT/multirank.log )
grep -h "FAILED\|FAIL " $OUT/*.log | head -20
