#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" [bench args]   (one rocprofv3 --pmc pass)
set -u
TAG=$1; CTRS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/pmc" -o run -- \
    python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-host-api $* > "$OUT/bench_pmc.log" 2>&1
python3 - "$OUT/pmc" <<'PY' | tee -a "$OUT/pmc_summary.txt"
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        if "csx" not in k: continue
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    for c, (v, n) in sorted(d.items()):
        print("%-40s %-28s per_launch=%.1f (n=%d)" % (k, c, v / max(n, 1), n))
PY
rm -rf "$OUT/pmc"
