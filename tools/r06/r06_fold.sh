#!/bin/bash
# round 6: the init pass folded into the launch (spx.gpu.init_fold) -- parity first (bounded by a timeout: the
# row-blocks' workgroups wait for the init workgroups), then on / off on the three workloads with an adding kernel
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06f; mkdir -p $OUT; cd $ROOT
timeout 900 python3 -m pytest tests/test_gpu_init_fold.py -x -q > $OUT/pytest_fold.txt 2>&1; tail -5 $OUT/pytest_fold.txt
R=$OUT/init_fold_ab.md; : > $R
SETS="off:spx.gpu.init_fold=false on:spx.gpu.init_fold=true auto: off2:spx.gpu.init_fold=false on2:spx.gpu.init_fold=true"
timeout 600 python3 tools/abl.py syn-nd24k --symmetric --steps 300 --header $SETS 2>$OUT/a.err | tee -a $R
timeout 600 python3 tools/abl.py syn-webbase --steps 300 $SETS 2>>$OUT/a.err | tee -a $R
timeout 600 python3 tools/abl.py syn-cant --symmetric --steps 300 $SETS 2>>$OUT/a.err | tee -a $R
timeout 900 python3 tools/abl.py syn-nlpkkt --edge 240 --symmetric --steps 50 $SETS 2>>$OUT/a.err | tee -a $R
