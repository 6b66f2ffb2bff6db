#!/bin/bash
# round-4 probe 20: the one test that takes 11 s with huge-page blocks and 4 s without (tests/test_gpu_parity.py::test_product_in_parts[2]), taken apart
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r04t; mkdir -p $OUT; cd $ROOT
python3 tools/r04/r04_probe20.py 2>&1 | grep "==\|INFO" | grep -v "Format\|Selected\|Encod" | tee $OUT/on.txt
SPX_NO_HUGE_PAGES=1 python3 tools/r04/r04_probe20.py 2>&1 | grep "==" | tee $OUT/off.txt
