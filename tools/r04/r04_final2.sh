#!/bin/bash
# round 4: what the driver runs at round end, on the final tree (whole GPU suite in one go, smoke, the default bench line),
# then a second soak on other seeds
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r04_final2; mkdir -p $OUT; cd $ROOT
( time python3 -m pytest tests -x -q -m gpu -p no:cacheprovider ) 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8 | tee $OUT/pytest_gpu_all.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
( time python3 bench.py ) > $OUT/bench_default.log 2>&1
grep '^{"metric"' $OUT/bench_default.log | tail -1 > $OUT/bench_default.json
tail -4 $OUT/bench_default.log | cut -c1-300
sed -i 's/20000 + k \* 1000)) \$((21000 + k \* 1000))/30000 + k * 1000)) $((31000 + k * 1000))/; s/4000 + k \* 250)) \$((4250 + k \* 250))/7000 + k * 250)) $((7250 + k * 250))/; s/5000 5200 --roundtrip/8000 8200 --roundtrip/; s/soak_rect.py 6000 7200/soak_rect.py 9000 10200/; s#gpurun_out/r04_soak#gpurun_out/r04_soak2#' tools/r04/r04_soak.sh
bash tools/r04/r04_soak.sh 2>&1 | tail -14
git -C $ROOT checkout tools/r04/r04_soak.sh 2>/dev/null
