#!/bin/bash
# round-3 probe 8: run-to-run spread of the bench matrix's product in separate processes; row-blocks dealt round robin
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r03i; mkdir -p $OUT; cd $ROOT
line() { python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1])
print('  %8.1f GF/s %8.4f ms  frac %.4f  read peak %.0f  batches %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('measured_stream_read_peak', 0), d['protocol']['batch_ms']))"; }
R=$OUT/spread.txt; : > $R
for rep in 1 2 3 4; do
    unset SPX_XCD_INTERLEAVE
    echo "== general e240 contiguous parts per XCD (rep $rep)" >> $R
    python3 bench.py --no-cpu-baseline --no-configs --steps 30 --warmup 10 2>/dev/null | line >> $R
    export SPX_XCD_INTERLEAVE=1
    echo "== general e240 row-blocks dealt round robin (rep $rep)" >> $R
    python3 bench.py --no-cpu-baseline --no-configs --steps 30 --warmup 10 2>/dev/null | line >> $R
done
unset SPX_XCD_INTERLEAVE
rocm-smi --showclocks --showpower 2>/dev/null | head -30 >> $R
cat $R
