#!/bin/bash
# round 6: FETCH_SIZE per dispatch of the unit-window kernel on joined row-blocks (the evidence run's average,
# 1.47 M KiB, lies below the bytes the kernel must read): distribution over the dispatches, both emissions pinned,
# plus the request-size counters behind the derived metric
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06fj; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
J="--opt spx.gpu.unit_windows=true --opt spx.gpu.rowblock_rows=1024 --opt spx.gpu.rowblock_elems=24576 --opt spx.gpu.unit_window_doubles=8192 --opt spx.gpu.waves=8"
P="--opt spx.gpu.unit_windows=true --opt spx.gpu.rowblock_elems=8192 --opt spx.gpu.waves=4"
for tag in joined plain; do
  [ $tag = joined ] && O="$J" || O="$P"
  for C in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    c1=$(echo $C | cut -d' ' -f1)
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${tag}_$c1 -o run -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs $O > $OUT/log_${tag}_$c1.txt 2>&1
    python3 - $OUT/pmc_${tag}_$c1 $tag >> $OUT/fetch_joined.txt <<'PY'
import csv, glob, sys, collections, statistics
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "csx_spmv" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:44], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    v.sort()
    print("%-7s %-44s %-22s n=%4d min=%.0f median=%.0f max=%.0f" % (sys.argv[2], k, c, len(v), v[0], statistics.median(v), v[-1]))
PY
    rm -rf $OUT/pmc_${tag}_$c1
  done
done
cat $OUT/fetch_joined.txt
