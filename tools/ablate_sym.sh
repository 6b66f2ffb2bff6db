#!/bin/bash
# Where the symmetric tile kernel's time goes (syn-nd24k --symmetric): rocprofv3 kernel
# durations of the full build and of variants with parts compiled out
# (tools/build_variant.sh SYM_x "-DSPX_ABL_SYM_x"; results are wrong on purpose).
#   SYM_VALSONLY  tile passes stream their descriptors and values only
#   SYM_NOSHFL    no exchange of the transposed products among a tile's eight lanes
#   SYM_NOATOMIC  no LDS adds (row sums, column sums)
#   SYM_NOSPILL   the column sums are not written to the spill array
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
export SPX_BENCH_ABLATION=1
for v in ${VARIANTS:-FULL SYM_VALSONLY SYM_NOSHFL SYM_NOATOMIC SYM_NOSPILL}; do
    if [ $v = FULL ]; then unset SPX_LIB_PATH; else export SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_$v.so; fi
    OUT=$ROOT/gpurun_out/abls_$v
    rm -rf $OUT; mkdir -p $OUT
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 $ROOT/bench.py --no-cpu-baseline --no-configs --steps 100 --warmup 10 --workload ${WORKLOAD:-syn-nd24k} --symmetric "$@" > $OUT/log 2>&1
    echo "== $v"
    python3 - $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "csx_" in r["Name"]:
            print("  %-70s calls %6s avg %9.1f ns  min %8s" % (r["Name"].split("(")[0][-70:], r["Calls"], float(r["AverageNs"]), r["MinNs"]))
PY
    rm -rf $OUT
done
