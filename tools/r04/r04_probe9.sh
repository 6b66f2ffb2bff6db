#!/bin/bash
# round-4 probe 9: the Cuthill-McKee order itself against the owner form on more ranks (kernel cost of the numbering),
# and the product in parts once more (the cut is cached now)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r04i; mkdir -p $OUT; cd $ROOT
S=$OUT/slices2_raw.md; : > $S
python3 tools/slice_time.py 8 --edge 240 --ranks 0,3,7 --reorder rcm --header 2>/dev/null | tee -a $S
python3 tools/slice_time.py 8 --edge 240 --ranks 0,3,7 --reorder rcm_owner 2>/dev/null | tee -a $S
python3 tools/slice_time.py 8 --edge 240 --ranks 3 --parts 2 2>/dev/null | tee -a $S
python3 tools/slice_time.py 8 --edge 240 --ranks 3 --parts 8 2>/dev/null | tee -a $S
python3 tools/slice_time.py 2 --edge 240 --ranks 0,1 --reorder rcm 2>/dev/null | tee -a $S
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "parts" 2>&1 | tail -2
