#!/bin/bash
# round-4 probe 7: whole GPU suite (incl. the new Matrix Market and parts tests), per-row modes A/B on webbase,
# one-GPU proxies of the multi-GPU step (slices timed one at a time; natural numbering vs spx_hip_dist_reorder)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r04g; mkdir -p $OUT; cd $ROOT
export PYTHONFAULTHANDLER=1
timeout 1800 python3 -X faulthandler -m pytest tests -x -q -m gpu -k "not multirank" -p no:cacheprovider --durations=8 > $OUT/pytest_gpu_full.txt 2>&1
tail -14 $OUT/pytest_gpu_full.txt | cut -c1-200
R=$OUT/webbase_row_modes_raw.md; : > $R
for i in 1 2; do
python3 tools/abl.py syn-webbase --steps 400 --header "row modes:" "row modes:" 2>/dev/null | tee -a $R
SPX_NO_ROW_MODES=1 python3 tools/abl.py syn-webbase --steps 400 "all rows added (SPX_NO_ROW_MODES):" "all rows added (SPX_NO_ROW_MODES):" 2>/dev/null | tee -a $R
done
S=$OUT/slices_raw.md; : > $S
python3 tools/slice_time.py 8 --edge 240 --ranks 0,3,7 --header 2>/dev/null | tee -a $S
python3 tools/slice_time.py 8 --edge 240 --ranks 0,3,7 --reorder rcm_owner 2>/dev/null | tee -a $S
python3 tools/slice_time.py 8 --edge 240 --ranks 3 --reorder rcm 2>/dev/null | tee -a $S
python3 tools/slice_time.py 2 --edge 240 --ranks 0,1 2>/dev/null | tee -a $S
python3 tools/slice_time.py 2 --edge 240 --ranks 0,1 --reorder rcm_owner 2>/dev/null | tee -a $S
python3 tools/slice_time.py 4 --edge 240 --ranks 1,2 --reorder rcm_owner 2>/dev/null | tee -a $S
python3 tools/slice_time.py 8 --edge 240 --ranks 1,4 --reorder rcm_owner --symmetric 2>/dev/null | tee -a $S
