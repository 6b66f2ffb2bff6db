#!/bin/bash
# round-5 probe 9: persistent workgroups (csx_spmv_xwp_kernel): parity, then A/B on the bench matrix
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05i; mkdir -p $OUT; cd $ROOT
timeout 1200 python3 -m pytest tests/test_gpu_unit_windows.py -q -x 2>&1 | tail -30 | tee $OUT/pytest.txt
grep -q passed $OUT/pytest.txt && ! grep -q failed $OUT/pytest.txt || exit 1
R=$OUT/abl.md; : > $R
X="spx.gpu.unit_windows=true,spx.gpu.unit_window_doubles=3072"
P="$X,spx.gpu.persistent=true"
SETS="off-w4:spx.gpu.unit_windows=false,spx.gpu.waves=4 xw-w4:$X,spx.gpu.waves=4"
for d in 2 3 4; do SETS="$SETS p4-d$d:$P,spx.gpu.unit_window_depth=$d"; done
for d in 2 3 4; do SETS="$SETS p8-d$d:$P,spx.gpu.persistent_waves=8,spx.gpu.unit_window_depth=$d"; done
for g in 1 2 3; do SETS="$SETS p4-d4-g$g:$P,spx.gpu.unit_window_depth=4,spx.gpu.persistent_wgs=$g"; done
timeout 1800 python3 tools/abl.py syn-nlpkkt --edge 240 --steps 40 --header $SETS 2>$OUT/abl240.err | tee -a $R
tail -3 $OUT/abl240.err
