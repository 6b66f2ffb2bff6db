#!/bin/bash
# round-4 probe 14: GPU suite + default bench line after the tune-time work (parallel partition sizing, launch
# autotune with ~20 ms per timing, pinned staging buffers allocated on first use)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r04n; mkdir -p $OUT; cd $ROOT
timeout 1800 python3 -X faulthandler -m pytest tests -x -q -m gpu -k "not multirank" -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
tail -4 $OUT/pytest_gpu.txt | cut -c1-200
( time python3 bench.py ) > $OUT/bench_default.log 2>&1
grep '^{"metric"' $OUT/bench_default.log | tail -1 > $OUT/bench_default.json
python3 -c "
import json
d = json.load(open('$OUT/bench_default.json'))
print('value %.1f GFLOP/s, %.4f ms, frac %.4f, tune %.2f s, emit+upload+autotune %.2f s, waves %d' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['format']['tune_seconds'], d['format']['emit_upload_seconds'], d['format']['waves_per_workgroup']))
for k, v in d['configs'].items(): print(k, v['gflops'], v['us_per_spmv'], v['roofline']['kernel'][:70], v['tune_seconds'], v['emit_upload_seconds'])
"
grep "^real" $OUT/bench_default.log
