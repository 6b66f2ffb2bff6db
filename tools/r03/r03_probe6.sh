#!/bin/bash
# round-3 probe 6: what x costs the general kernel on the bench matrix (ablation build), load balance of eight slices
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r03g; mkdir -p $OUT; cd $ROOT
bash tools/build_variant.sh NOX "-DSPX_ABL_NOX" > /dev/null 2>&1
bash tools/build_variant.sh NOATOMIC "-DSPX_ABL_NOATOMIC" > /dev/null 2>&1
bash tools/build_variant.sh VALSONLY "-DSPX_ABL_VALSONLY" > /dev/null 2>&1
line() { python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1])
print('  %8.1f GF/s %8.4f ms  frac %.4f  %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel']))"; }
R=$OUT/general_ablation_e240.txt; : > $R
export SPX_BENCH_ABLATION=1
for v in FULL NOX NOATOMIC VALSONLY FULL; do
    if [ $v = FULL ]; then unset SPX_LIB_PATH; else export SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_$v.so; fi
    echo "== general e240 $v" >> $R
    python3 bench.py --no-cpu-baseline --no-configs --steps 50 --warmup 10 2>/dev/null | line >> $R
done
unset SPX_LIB_PATH SPX_BENCH_ABLATION
cat $R
python tools/slice_time.py 8 240 0,3,4,7 > $OUT/slices_world8_edge240.txt 2>$OUT/slices.err; cat $OUT/slices_world8_edge240.txt
python tools/slice_time.py 2 240 > $OUT/slices_world2_edge240.txt 2>>$OUT/slices.err; cat $OUT/slices_world2_edge240.txt
