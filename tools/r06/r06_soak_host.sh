#!/bin/bash
# round 6: soak of the host-vector entry point -- the larger random matrices of tools/soak_large.py, every product also
# through library vectors with the thresholds lowered so that x (and y) travel piece by piece in the order the parts
# need them (general and symmetric streams, all the round's options in the draws)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06s; mkdir -p $OUT; cd $ROOT
export SPX_HOST_PARTS_MIN_BYTES=1024 SPX_HOST_XPIECE_BYTES=16384
timeout 2400 python3 tools/soak_large.py ${1:-500} ${2:-620} --library-vectors > $OUT/soak_host.txt 2>&1
tail -1 $OUT/soak_host.txt; grep -c ' ok' $OUT/soak_host.txt; grep FAILED $OUT/soak_host.txt | head -5 | cut -c1-300
