#!/bin/bash
# per-kernel rocprofv3 durations of one bench.py configuration: tools/prof_kernels.sh <bench args...>
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/profk
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 $ROOT/bench.py --no-cpu-baseline --no-configs --steps 300 --warmup 30 "$@" > $OUT/log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-60s calls %5s avg %9.1f ns  min %8s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]), r["MinNs"]))
PY
