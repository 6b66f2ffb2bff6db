#!/bin/bash
# round-4 probe 1: run-to-run spread of the bench matrix's product -- fresh processes restoring the SAME saved
# stream, alternating one-arena / per-array allocation, plain and under rocprofv3 --pmc (translation counters);
# then the GPU suite on the arena build
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r04a; mkdir -p $OUT; cd $ROOT
F=/tmp/e240.spx
R=$OUT/spread.txt; : > $R
python3 tools/spread_probe.py save $F 2> $OUT/save.err | tee -a $R
for i in 1 2 3 4; do
    SPX_LOG_PLACEMENT=1 python3 tools/spread_probe.py time $F --tag arena$i 2>> $OUT/placement.txt | tee -a $R
    SPX_NO_ARENA=1 SPX_LOG_PLACEMENT=1 python3 tools/spread_probe.py time $F --tag split$i 2>> $OUT/placement.txt | tee -a $R
done
summ() { python3 - "$1" "$2" <<'PY' | tee -a $R
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0]); dur = [0.0, 0]; seen = set()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "csx_spmv_kernel" not in r["Kernel_Name"]: continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        key = r.get("Dispatch_Id")
        if key not in seen and r.get("Start_Timestamp"):
            seen.add(key); dur[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); dur[1] += 1
print("PMC %s: kernel avg %.1f us over %d dispatches; " % (sys.argv[2], 1e-3 * dur[0] / max(dur[1], 1), dur[1]) +
      "  ".join("%s=%.4g" % (c, v / max(n, 1)) for c, (v, n) in sorted(acc.items())))
PY
}
cd /tmp && export TMPDIR=/tmp
SETA="TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum"
SETB="TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_LFIFO_FULL_sum TCP_PENDING_STALL_CYCLES_sum GRBM_UTCL2_BUSY"
SETC="TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum TCC_TAG_STALL_sum TCP_TCR_TCP_STALL_CYCLES_sum"
for i in 1 2 3; do
  for S in A B C; do
    eval CTR=\$SET$S
    rm -rf $OUT/pmc
    rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT/pmc -o run -- \
        python3 $ROOT/tools/spread_probe.py time $F --steps 8 --batches 3 --tag pmc${S}_arena$i 2>> $OUT/pmc.err | grep '^{' | tee -a $R
    summ $OUT/pmc "set$S arena$i"
    rm -rf $OUT/pmc
    SPX_NO_ARENA=1 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT/pmc -o run -- \
        python3 $ROOT/tools/spread_probe.py time $F --steps 8 --batches 3 --tag pmc${S}_split$i 2>> $OUT/pmc.err | grep '^{' | tee -a $R
    summ $OUT/pmc "set$S split$i"
  done
done
rm -rf $OUT/pmc
cd $ROOT
rocm-smi --showclocks --showpower --showmeminfo vram 2>/dev/null | head -40 >> $R
timeout 900 python3 -m pytest tests -x -q -m gpu -k "not fullsize and not multirank" 2>&1 | tail -5 | tee $OUT/pytest_gpu.txt
