#!/bin/bash
# Profiles bench.py on the GPU box: per-kernel times (rocprofv3 --kernel-trace
# --stats) and, in separate runs, the HBM traffic counters.  Outputs land in
# gpurun_out/<tag>/; copy the summaries worth keeping into profiles/.
# usage: tools/profile.sh <tag> [bench.py args...]
set -u
TAG=${1:-prof}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-configs --no-host-api $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o run -- \
    python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_trace.log" 2>&1
grep '^{"metric"' "$OUT/bench_trace.log" | tail -1 > "$OUT/bench_line.json"
find "$OUT/trace" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
rm -rf "$OUT/trace"/*/*kernel_trace.csv "$OUT/trace"/*kernel_trace.csv 2>/dev/null
# counters: one pass each (FETCH_SIZE and WRITE_SIZE do not fit one pass)
for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -o run -- \
        python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-host-api $* > "$OUT/bench_pmc_$C.log" 2>&1
    python3 - "$OUT/pmc_$C" "$C" > "$OUT/pmc_$C.txt" <<'PY'
import csv, glob, sys, collections
d, name = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == name:
            k = r["Kernel_Name"][:60]
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for k, (v, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print("%-60s launches=%d %s_sum=%.0f per_launch=%.1f" % (k, n, name, v, v / max(n, 1)))
PY
    rm -rf "$OUT/pmc_$C"
done
cat "$OUT/kernel_stats.csv" | head -8
cat "$OUT"/pmc_*.txt | head -8
