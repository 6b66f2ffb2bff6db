#!/bin/bash
# round 6: syn-webbase, column slices launched in turn (slice 0 stores y, the others add; no scale kernel, no
# atomics) against the slices on groups of XCDs in one launch that the tuner measures (c2 / c4)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06wp; mkdir -p $OUT; cd $ROOT
R=$OUT/webbase_phases.md; : > $R
SETS="default: c2:spx.gpu.col_phases=c2 s2:spx.gpu.col_phases=2 s3:spx.gpu.col_phases=3 s4:spx.gpu.col_phases=4 s8:spx.gpu.col_phases=8 c1:spx.gpu.col_phases=1 default2:"
timeout 900 python3 tools/abl.py syn-webbase --steps 300 --header $SETS 2>$OUT/a.err | tee -a $R
