#!/bin/bash
# round 4, after the clean rebuild of the last tree: smoke, the parity files, and the rocprof summary + PMC passes of the bench command
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r04_last; mkdir -p $OUT; cd $ROOT
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_client.py tests/test_gpu_random.py -x -q -m gpu ) 2>&1 | tail -6 | tee $OUT/pytest_subset.txt
bash tools/profile.sh r04_last_prof 2>&1 | tail -12
