#!/bin/bash
# round-4 probe 4: passes in adjacent pairs + x loaded once for a couple (spx.gpu.pair_x): parity, then A/B in one process
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r04d; mkdir -p $OUT; cd $ROOT
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_launch_config.py tests/test_gpu_graph.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest_parity.txt
R=$OUT/pair_x_raw.md; : > $R
python3 tools/abl.py syn-nlpkkt --edge 240 --header --steps 60 "pair_x:" "no pair_x:spx.gpu.pair_x=false" "pair_x:" "no pair_x:spx.gpu.pair_x=false" 2>/dev/null | tee -a $R
python3 tools/abl.py syn-nlpkkt --edge 120 "pair_x:" "no pair_x:spx.gpu.pair_x=false" "pair_x:" "no pair_x:spx.gpu.pair_x=false" 2>/dev/null | tee -a $R
python3 tools/abl.py syn-kkt2f --edge 100 "pair_x:" "no pair_x:spx.gpu.pair_x=false" 2>/dev/null | tee -a $R
python3 tools/abl.py syn-cant --steps 400 "pair_x:" "no pair_x:spx.gpu.pair_x=false" 2>/dev/null | tee -a $R
python3 tools/abl.py syn-nd24k --steps 300 "pair_x:" "no pair_x:spx.gpu.pair_x=false" 2>/dev/null | tee -a $R
python3 tools/abl.py syn-nd24k --symmetric --steps 300 "default:" 2>/dev/null | tee -a $R
python3 tools/abl.py syn-webbase --steps 300 "default:" 2>/dev/null | tee -a $R
python3 tools/abl.py syn-nlpkkt --edge 240 --symmetric --steps 60 "default:" 2>/dev/null | tee -a $R
