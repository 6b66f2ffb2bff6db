#!/bin/bash
# round-4 probe 5: band launch order WITH the values moved along (one ascending stream of values per XCD): parity, A/B
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r04e; mkdir -p $OUT; cd $ROOT
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_get_set_entry.py tests/test_save_restore.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest_parity.txt
R=$OUT/band_order_values_raw.md; : > $R
python3 tools/abl.py syn-nlpkkt --edge 240 --header --steps 60 "stream order:" "band order:spx.gpu.band_order=true" "stream order:" "band order:spx.gpu.band_order=true" 2>/dev/null | tee -a $R
python3 tools/abl.py syn-nlpkkt --edge 120 "stream order:" "band order:spx.gpu.band_order=true" "stream order:" "band order:spx.gpu.band_order=true" 2>/dev/null | tee -a $R
python3 tools/abl.py syn-kkt2f --edge 100 "stream order:" "band order:spx.gpu.band_order=true" 2>/dev/null | tee -a $R
python3 tools/abl.py syn-nlpkkt --edge 240 --symmetric --steps 60 "stream order:" "band order:spx.gpu.band_order=true" 2>/dev/null | tee -a $R
