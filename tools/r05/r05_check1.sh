#!/bin/bash
# round-5 check 1: unit-window parity tests, then the default bench matrix with the launch tuner deciding
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05k; mkdir -p $OUT; cd $ROOT
timeout 1200 python3 -m pytest tests/test_gpu_unit_windows.py -q -x 2>&1 | tail -4 | tee $OUT/pytest.txt
timeout 900 python3 bench.py --no-cpu-baseline --no-configs 2>$OUT/bench.err | tee $OUT/bench.json | cut -c1-1500
timeout 900 python3 tools/abl.py syn-nlpkkt --edge 240 --steps 40 --header "off:spx.gpu.unit_windows=false" "auto:" "on:spx.gpu.unit_windows=true" 2>>$OUT/bench.err | tee $OUT/abl.md
