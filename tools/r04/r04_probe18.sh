#!/bin/bash
# round-4 probe 18: durations of the parity files of the GPU suite (a 161 s run of what took 35 s before needs explaining)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r04r; mkdir -p $OUT; cd $ROOT
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_client.py -x -q -m gpu --durations=12 ) 2>&1 | tail -25 | tee $OUT/durations.txt
( time SPX_NO_HUGE_PAGES=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_client.py -x -q -m gpu --durations=5 ) 2>&1 | tail -14 | tee $OUT/durations_no_huge.txt
