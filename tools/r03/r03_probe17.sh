#!/bin/bash
# round-3 probe 17: x of a row segment in pairs at any alignment (one 16-byte load per two columns), A/B builds on the box
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r03r; mkdir -p $OUT; cd $ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -x -q -n 4 > $OUT/pytest_parity.log 2>&1; tail -n 3 $OUT/pytest_parity.log
bash tools/build_variant.sh ALIGNEDONLY "-DSPX_X_PAIRS_ALIGNED_ONLY" > /dev/null 2>&1
line() { python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1])
print('  %8.1f GF/s %8.4f ms  frac %.4f  %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel']))"; }
R=$OUT/ab_xpairs.txt; : > $R
for rep in 1 2 3 4; do for v in FULL ALIGNEDONLY; do
    if [ $v = FULL ]; then unset SPX_LIB_PATH; else export SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_$v.so; fi
    echo "== general e120 $v (rep $rep)" >> $R
    python3 bench.py --no-cpu-baseline --no-configs --steps 100 --warmup 10 --edge 120 --opt spx.gpu.wave_tiles=false 2>/dev/null | line >> $R
done; done
for rep in 1 2 3; do for v in FULL ALIGNEDONLY; do
    if [ $v = FULL ]; then unset SPX_LIB_PATH; else export SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_$v.so; fi
    echo "== symmetric e120 $v (rep $rep)" >> $R
    python3 bench.py --no-cpu-baseline --no-configs --steps 100 --warmup 10 --edge 120 --symmetric 2>/dev/null | line >> $R
done; done
for rep in 1 2; do for v in FULL ALIGNEDONLY; do
    if [ $v = FULL ]; then unset SPX_LIB_PATH; else export SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_$v.so; fi
    echo "== general e240 $v (rep $rep)" >> $R
    python3 bench.py --no-cpu-baseline --no-configs --steps 50 --warmup 10 --opt spx.gpu.wave_tiles=false 2>/dev/null | line >> $R
done; done
for w in syn-cant "syn-kkt2f --edge 100"; do for v in FULL ALIGNEDONLY FULL ALIGNEDONLY; do
    if [ $v = FULL ]; then unset SPX_LIB_PATH; else export SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_$v.so; fi
    echo "== $w $v" >> $R
    python3 bench.py --no-cpu-baseline --no-configs --steps 200 --warmup 30 --workload $w 2>/dev/null | line >> $R
done; done
unset SPX_LIB_PATH
cat $R
