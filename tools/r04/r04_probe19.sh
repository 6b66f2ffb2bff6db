#!/bin/bash
# round-4 probe 19: why the parity files took 10 min after the two contract-size tunes (35 s on a fresh box): the same sequence with
# durations, the load and the kernel's huge-page / compaction counters before and after
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r04s; mkdir -p $OUT; cd $ROOT
grep -i "thp_fault\|thp_collapse\|compact_stall\|compact_fail\|compact_success\|thp_split_page " /proc/vmstat > $OUT/vmstat_before.txt; uptime
bash tools/r04/r04_probe16.sh 2>&1 | tail -22
uptime
grep -i "thp_fault\|thp_collapse\|compact_stall\|compact_fail\|compact_success\|thp_split_page " /proc/vmstat > $OUT/vmstat_after.txt
paste $OUT/vmstat_before.txt $OUT/vmstat_after.txt
