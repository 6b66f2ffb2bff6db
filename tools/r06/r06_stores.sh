#!/bin/bash
# round 6, VERDICT r05 item 3(a): the store flavours not yet tried for the write-out of y -- `sc1` and `sc0 sc1` (they
# drop the line from the XCD's L2), 8 and 16 bytes per lane -- and atomic adds, in the pattern probe
# (tools/micro/stream_pattern.hip mode 6: a 64 KB chunk read per workgroup, 2.4 KB written at its end)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06st; mkdir -p $OUT; cd $ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/stream_pattern.hip -o $OUT/stream_pattern 2>/dev/null
timeout 600 $OUT/stream_pattern 64 40 > $OUT/pattern_store_flavours.txt 2>&1
grep "mode 6" $OUT/pattern_store_flavours.txt
