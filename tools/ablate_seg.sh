#!/bin/bash
# Where the read-once segment kernel's time goes (syn-nlpkkt --symmetric): launch period of the
# full build and of variants with parts compiled out (results wrong on purpose).
#   SEG_NOSLOTADD  the transposed products are not added to the LDS slots
#   SEG_NOFLUSH    the slots are not handed to y (no global atomics for them)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export SPX_BENCH_ABLATION=1
for v in ${VARIANTS:-FULL SEG_NOSLOTADD SEG_NOFLUSH SEG_NEITHER}; do
    if [ $v = FULL ]; then unset SPX_LIB_PATH; else export SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_$v.so; fi
    echo "== $v"
    python3 $ROOT/bench.py --no-cpu-baseline --no-configs --steps 100 --warmup 10 --edge ${EDGE:-120} --symmetric --opt spx.gpu.sym_segments=true "$@" 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('  %8.1f GF/s %8.4f ms  rb %d W%d' % (d['value'], d['ms_per_step'], d['format']['rowblocks'], d['format']['waves_per_workgroup']))"
done
