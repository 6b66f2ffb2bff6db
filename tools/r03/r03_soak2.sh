#!/bin/bash
# round 3: the long soak (the round's options in every draw)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r03_soak2; mkdir -p $OUT; cd $ROOT
python -m pytest tests/test_gpu_parity.py -q -x -k "band_launch_order" > $OUT/pytest_band.log 2>&1; tail -n 3 $OUT/pytest_band.log
for k in 0 1 2 3; do ( timeout 1500 python tools/soak_random.py $((1400 + k * 1200)) $((2600 + k * 1200)) > $OUT/random_$k.log 2>&1; tail -n 2 $OUT/random_$k.log ) & done
for k in 0 1 2; do ( timeout 1500 python tools/soak_large.py $((300 + k * 300)) $((600 + k * 300)) > $OUT/large_$k.log 2>&1; tail -n 1 $OUT/large_$k.log ) & done
( timeout 1500 python tools/soak_large.py 1080 1300 --roundtrip > $OUT/roundtrip.log 2>&1; tail -n 1 $OUT/roundtrip.log ) &
( timeout 1500 python tools/soak_rect.py 0 1500 > $OUT/rect.log 2>&1; tail -n 2 $OUT/rect.log ) &
wait
( timeout 1500 bash tools/soak_multirank.sh > $OUT/multirank.log 2>&1; tail -n 3 $OUT/multirank.log; grep -c "^ok" $OUT/multirank.log )
grep -h "FAILED\|FAIL " $OUT/*.log | head -20
