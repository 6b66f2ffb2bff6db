#!/bin/bash
# round-4 probe 10: single-descriptor passes with the descriptor folded onto lane 0 (row = r0 + l * drow, col = c0 + l * dcol:
# no kind / step / segment-number decode in the wavefront), A/B in one process
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r04j; mkdir -p $OUT; cd $ROOT
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "folded or parts" 2>&1 | tail -3 | tee $OUT/pytest.txt
R=$OUT/folded_desc_raw.md; : > $R
python3 tools/abl.py syn-nlpkkt --edge 240 --header --steps 60 "inline:" "folded:spx.gpu.inline_desc=folded" "inline:" "folded:spx.gpu.inline_desc=folded" "loaded:spx.gpu.inline_desc=false" 2>/dev/null | tee -a $R
python3 tools/abl.py syn-nlpkkt --edge 120 "inline:" "folded:spx.gpu.inline_desc=folded" "inline:" "folded:spx.gpu.inline_desc=folded" 2>/dev/null | tee -a $R
python3 tools/abl.py syn-kkt2f --edge 100 "inline:" "folded:spx.gpu.inline_desc=folded" 2>/dev/null | tee -a $R
python3 tools/abl.py syn-cant --steps 400 "inline:" "folded:spx.gpu.inline_desc=folded" "inline:" "folded:spx.gpu.inline_desc=folded" 2>/dev/null | tee -a $R
python3 tools/abl.py syn-nd24k --steps 300 "inline:" "folded:spx.gpu.inline_desc=folded" 2>/dev/null | tee -a $R
