#!/bin/bash
# Where the kernel's time goes: launch-time (HIP events) of the full kernel and of
# variants with parts compiled out (tools/build_variant.sh; results are wrong on
# purpose, the parity gate is skipped and the lines are marked invalid).
#   EMPTY     every workgroup returns at once: launch + dispatch floor
#   NOPASS    row-block header, tile init, write-out; no passes
#   VALSONLY  pass headers, descriptors and values streamed; no x, no LDS adds
#   NOX       everything but the x gathers
#   NOATOMIC  everything but the LDS adds
cd "$(dirname "$0")/.."
for w in ${WORKLOADS:-syn-cant syn-nd24k syn-webbase}; do
  for v in FULL EMPTY NOPASS VALSONLY NOX NOATOMIC; do
    if [ $v = FULL ]; then unset SPX_LIB_PATH; else export SPX_LIB_PATH=$PWD/sparsex_amd/lib/variants/libsparsex_$v.so; fi
    SPX_BENCH_ABLATION=1 python bench.py --no-cpu-baseline --no-configs --steps 400 --workload $w "$@" 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('%-12s %-9s launch %7.2f us   step %7.2f us' % (d['config']['workload'][:11], '$v', d['roofline']['avg_launch_us'], d['ms_per_step'] * 1e3))"
  done
done
