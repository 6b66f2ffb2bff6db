#!/bin/bash
# round-5 probe 10: persistent workgroups with a loader wavefront: geometry x depth x LDS budget
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05j; mkdir -p $OUT; cd $ROOT
timeout 1200 python3 -m pytest tests/test_gpu_unit_windows.py -q -x 2>&1 | tail -4 | tee $OUT/pytest.txt
grep -q passed $OUT/pytest.txt && ! grep -q failed $OUT/pytest.txt || exit 1
R=$OUT/abl.md; : > $R
SETS="off-w4:spx.gpu.unit_windows=false,spx.gpu.waves=4"
for b in 3072 2688; do
  P="spx.gpu.unit_windows=true,spx.gpu.unit_window_doubles=$b,spx.gpu.persistent=true"
  for w in 4 7; do for d in 2 3 4; do for g in 2 3; do
    SETS="$SETS b$b-w$w-d$d-g$g:$P,spx.gpu.persistent_waves=$w,spx.gpu.unit_window_depth=$d,spx.gpu.persistent_wgs=$g"
  done; done; done
done
timeout 2400 python3 tools/abl.py syn-nlpkkt --edge 240 --steps 40 --header $SETS 2>$OUT/abl240.err | tee -a $R
tail -3 $OUT/abl240.err
