#!/bin/bash
# round-4 probe 3: (a) does the product's level follow the physical placement? (one process, the stream uploaded
# eight times with other allocations kept in between); (b) the multi-rank tests with the halo exchange, the
# overlapped step and the renumbered matrices; (c) the contract matrix on eight ranks, three ways
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r04c; mkdir -p $OUT; cd $ROOT
F=/tmp/e240.spx
python3 tools/spread_probe.py save $F 2> $OUT/save.err > $OUT/save.txt
SPX_LOG_PLACEMENT=1 python3 tools/spread_probe.py place $F --steps 40 2> $OUT/place.err | tee $OUT/place.txt
SPX_LOG_PLACEMENT=1 python3 tools/spread_probe.py place $F --steps 40 2>> $OUT/place.err | tee -a $OUT/place.txt
rm -f $F
timeout 1800 python3 -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "not eight" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -15 | tee $OUT/pytest_multirank.txt
timeout 3000 python3 -m pytest tests/test_gpu_multirank.py -q -m gpu -k "eight" --durations=5 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -40 | tee $OUT/pytest_eight.txt
