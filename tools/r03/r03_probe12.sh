#!/bin/bash
# round-3 probe 12: is x a bandwidth or an instruction / latency cost?  (ablation: the same x loads out of 8 KB)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r03m; mkdir -p $OUT; cd $ROOT
bash tools/build_variant.sh XSMALL "-DSPX_ABL_XSMALL" > /dev/null 2>&1
bash tools/build_variant.sh NOX "-DSPX_ABL_NOX" > /dev/null 2>&1
line() { python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1])
print('  %8.1f GF/s %8.4f ms  frac %.4f  %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel']))"; }
R=$OUT/x_cost.txt; : > $R
export SPX_BENCH_ABLATION=1
for v in FULL XSMALL NOX FULL XSMALL NOX; do
    if [ $v = FULL ]; then unset SPX_LIB_PATH; else export SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_$v.so; fi
    echo "== general e240 $v" >> $R
    python3 bench.py --no-cpu-baseline --no-configs --steps 50 --warmup 10 2>/dev/null | line >> $R
done
cat $R
