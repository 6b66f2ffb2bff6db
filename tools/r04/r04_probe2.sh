#!/bin/bash
# round-4 probe 2: (a) host memory of the box; (b) the product's time as a series inside ONE process next to the
# clocks / power the driver reports (sampled by this shell, 5 Hz: sysfs) -- does the level move inside a process?
# (c) multi-rank tests incl. the contract matrix on eight ranks sharing the GPU; (d) the bench-size pytest case
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r04b; mkdir -p $OUT; cd $ROOT
{ free -g; cat /sys/fs/cgroup/memory.max 2>/dev/null; cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc; df -h /tmp | tail -1; } > $OUT/host.txt 2>&1
F=/tmp/e240.spx
python3 tools/spread_probe.py save $F 2> $OUT/save.err > $OUT/save.txt
DEV=$(ls -d /sys/class/drm/card*/device 2>/dev/null | head -1)
ls $DEV > $OUT/sysfs_ls.txt 2>&1
sample() {
    while true; do
        T=$(date +%s.%N)
        S=$(grep '\*' $DEV/pp_dpm_sclk 2>/dev/null | tr -d '\n'); M=$(grep '\*' $DEV/pp_dpm_mclk 2>/dev/null | tr -d '\n')
        Fc=$(grep '\*' $DEV/pp_dpm_fclk 2>/dev/null | tr -d '\n'); So=$(grep '\*' $DEV/pp_dpm_socclk 2>/dev/null | tr -d '\n')
        P=$(cat $DEV/hwmon/hwmon*/power1_average 2>/dev/null | head -1); if [ -z "$P" ]; then P=$(cat $DEV/hwmon/hwmon*/power1_input 2>/dev/null | head -1); fi
        TE=$(cat $DEV/hwmon/hwmon*/temp*_input 2>/dev/null | tr '\n' ' ')
        echo "CLK $T sclk[$S] mclk[$M] fclk[$Fc] soc[$So] power[$P] temps[$TE]"
        sleep 0.2
    done
}
sample > $OUT/clocks.txt 2>&1 &
SPID=$!
( while true; do date +%s.%N; rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "clock level\|Power\|Temperature"; sleep 2; done ) > $OUT/smi.txt 2>&1 &
SPID2=$!
python3 tools/spread_probe.py soak $F --seconds 70 --steps 40 --gap 3 --gap-every 200 --tag soak1 2> $OUT/soak1.err > $OUT/soak1.txt
python3 tools/spread_probe.py soak $F --seconds 40 --steps 40 --tag soak2 2> $OUT/soak2.err > $OUT/soak2.txt
kill $SPID $SPID2
python3 - $OUT <<'PY'
import sys
out = sys.argv[1]
for f in ("soak1.txt", "soak2.txt"):
    v = [float(l.split()[2]) for l in open(out + "/" + f) if l.startswith("SOAK")]
    if v:
        print(f, "batches", len(v), "min %.1f max %.1f first5 %s last5 %s" % (min(v), max(v), v[:5], v[-5:]))
PY
rm -f $F
timeout 1500 python3 -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "not eight" 2>&1 | tail -5 | tee $OUT/pytest_multirank.txt
AVAIL=$(python3 - <<'PY'
import re
m = int(re.search(r"MemAvailable:\s+(\d+)", open("/proc/meminfo").read()).group(1)) // (1 << 20)
try:
    c = open("/sys/fs/cgroup/memory.max").read().strip()
    if c != "max":
        m = min(m, int(c) >> 30)
except OSError:
    pass
print(m)
PY
)
echo "memory available to this container: $AVAIL GB" | tee -a $OUT/host.txt
if [ "$AVAIL" -ge 120 ]; then
    ( while true; do free -g | sed -n 2p; sleep 5; done ) > $OUT/mem_during_eight.txt 2>&1 &
    MPID=$!
    timeout 3000 python3 -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "eight" --durations=5 2>&1 | tail -25 | tee $OUT/pytest_eight.txt
    kill $MPID
else
    echo "SKIPPED: eight ranks of the contract matrix need ~80 GB of host memory" | tee $OUT/pytest_eight.txt
fi
free -g >> $OUT/host.txt
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "e240" --durations=5 2>&1 | tail -8 | tee $OUT/pytest_e240.txt
