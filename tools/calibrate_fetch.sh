#!/bin/bash
# What rocprofv3's FETCH_SIZE counts per scattered 8-byte read on gfx950: the gather
# microbenchmark over an array far beyond L2 + Infinity Cache (every gather a miss of known
# count) and over 8 MB (syn-webbase's x), one --pmc pass each.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
hipcc -w --offload-arch=gfx950 -O3 -munsafe-fp-atomics $ROOT/tools/micro/gather_atomic.hip -o /tmp/ga || exit 1
cd /tmp && export TMPDIR=/tmp
for m in "4096 4000000" "8 2500000" "8 3600000"; do
    rm -rf /tmp/pm
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pm -o run -- /tmp/ga gather $m > /tmp/pm.log 2>&1
    python3 - "$m" <<'PY'
import csv, glob, sys
mb, n = sys.argv[1].split()
vals = []
for f in glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == "FETCH_SIZE" and "gather_kernel" in r["Kernel_Name"]:
            vals.append(float(r["Counter_Value"]))
if vals:
    avg = sum(vals) / len(vals)
    print("gather %s MB, %s gathers: FETCH_SIZE %.0f KiB per launch (%d launches) = %.1f bytes per gather as reported, "
          "%.1f doubled per the guide's wide-read correction; index stream %.1f KiB"
          % (mb, n, avg, len(vals), avg * 1024 / int(n), 2 * avg * 1024 / int(n), int(n) * 4 / 1024))
else:
    print("gather %s: no FETCH_SIZE rows" % sys.argv[1])
PY
done
