#!/bin/bash
# round 4, the very last tree: what the driver runs at round end (whole GPU suite, smoke, default bench line)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r04_final9; mkdir -p $OUT; cd $ROOT
( time python3 -m pytest tests -x -q -m gpu -p no:cacheprovider ) 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8 | tee $OUT/pytest_gpu_all.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
( time python3 bench.py ) > $OUT/bench_default.log 2>&1
grep '^{"metric"' $OUT/bench_default.log | tail -1 > $OUT/bench_default.json
python3 -c "
import json
d = json.load(open('$OUT/bench_default.json'))
print('value %.1f GFLOP/s, %.4f ms, frac %.4f, tune %.2f s, emit+upload+autotune %.2f s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['format']['tune_seconds'], d['format']['emit_upload_seconds']))
"
grep "^real" $OUT/bench_default.log
