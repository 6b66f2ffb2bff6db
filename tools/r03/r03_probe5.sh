#!/bin/bash
# round-3 probe 5: non-temporal loads of the matrix stream, two read-once passes side by side (A/B builds made on the box)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r03f; mkdir -p $OUT; cd $ROOT
bash tools/build_variant.sh TEMPORAL "-DSPX_STREAM_TEMPORAL" > /dev/null 2>&1
bash tools/build_variant.sh SINGLE "-DSPX_SYMSEG_SINGLE" > /dev/null 2>&1
bash tools/build_variant.sh BOTHOLD "-DSPX_SYMSEG_SINGLE -DSPX_STREAM_TEMPORAL" > /dev/null 2>&1
line() { python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1])
print('  %8.1f GF/s %8.4f ms  frac %.4f  %s  modes %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'], d['protocol']['launch_modes']))"; }
R=$OUT/ab.txt; : > $R
for rep in 1 2; do
for v in FULL TEMPORAL; do
    if [ $v = FULL ]; then unset SPX_LIB_PATH; else export SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_$v.so; fi
    echo "== general e240 $v (rep $rep)" >> $R
    python3 bench.py --no-cpu-baseline --no-configs --steps 50 --warmup 10 2>/dev/null | line >> $R
done
done
for v in FULL SINGLE TEMPORAL BOTHOLD FULL; do
    if [ $v = FULL ]; then unset SPX_LIB_PATH; else export SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_$v.so; fi
    echo "== symmetric e240 $v" >> $R
    python3 bench.py --no-cpu-baseline --no-configs --steps 50 --warmup 10 --symmetric 2>/dev/null | line >> $R
done
unset SPX_LIB_PATH
for w in syn-cant syn-webbase; do for v in FULL TEMPORAL; do
    if [ $v = FULL ]; then unset SPX_LIB_PATH; else export SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_$v.so; fi
    echo "== $w $v" >> $R
    python3 bench.py --no-cpu-baseline --no-configs --steps 300 --warmup 30 --workload $w 2>/dev/null | line >> $R
done; done
for v in FULL TEMPORAL SINGLE; do
    if [ $v = FULL ]; then unset SPX_LIB_PATH; else export SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_$v.so; fi
    echo "== syn-nd24k symmetric $v" >> $R
    python3 bench.py --no-cpu-baseline --no-configs --steps 300 --warmup 30 --workload syn-nd24k --symmetric 2>/dev/null | line >> $R
    echo "== syn-kkt2f e120 symmetric $v" >> $R
    python3 bench.py --no-cpu-baseline --no-configs --steps 100 --warmup 10 --workload syn-kkt2f --edge 120 --symmetric 2>/dev/null | line >> $R
done
unset SPX_LIB_PATH
cat $R
(time python -m pytest tests -m gpu -x -q -n 4) > $OUT/pytest.log 2>&1; tail -n 4 $OUT/pytest.log
