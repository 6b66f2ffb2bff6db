#!/bin/bash
# round-3 probe 10: where the read-once kernel's stream time goes (more ablation builds); HBM temperature next to the run-to-run spread
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r03k; mkdir -p $OUT; cd $ROOT
for v in SEG_VALSONLY SEG_NOX SEG_NOFLUSH; do bash tools/build_variant.sh $v "-DSPX_ABL_$v" > /dev/null 2>&1; done
bash tools/build_variant.sh SEG_NOX_NOFLUSH "-DSPX_ABL_SEG_NOX -DSPX_ABL_SEG_NOFLUSH" > /dev/null 2>&1
VARIANTS="FULL SEG_VALSONLY SEG_NOX SEG_NOFLUSH SEG_NOX_NOFLUSH FULL" EDGE=240 bash tools/ablate_seg.sh > $OUT/symseg_ablation2_e240.txt 2>&1
cat $OUT/symseg_ablation2_e240.txt
line() { python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1])
print('  %8.1f GF/s %8.4f ms  frac %.4f  read peak %.0f' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('measured_stream_read_peak', 0)))"; }
R=$OUT/spread_temp.txt; : > $R
for rep in 1 2 3 4 5; do
    echo "== rep $rep, before:" >> $R; rocm-smi --showtemp --showclocks --showpower 2>/dev/null | grep -E "Temperature|mclk|fclk|sclk|Power" >> $R
    python3 bench.py --no-cpu-baseline --no-configs --steps 30 --warmup 10 2>/dev/null | line >> $R
    echo "   after:" >> $R; rocm-smi --showtemp 2>/dev/null | grep -E "Temperature" >> $R
    if [ $rep = 3 ]; then sleep 60; echo "   (slept 60 s)" >> $R; fi
done
cat $R
