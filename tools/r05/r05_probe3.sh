#!/bin/bash
# round-5 probe 3: second form of the unit-window kernel (pipeline stage for any width <= 4, pass headers in
# LDS, first round's loads in front of the barrier): parity tests, A/B, ablation builds.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05c; mkdir -p $OUT; cd $ROOT
timeout 1200 python3 -m pytest tests/test_gpu_unit_windows.py -q -x 2>&1 | tail -25 > $OUT/pytest.txt
cat $OUT/pytest.txt
R=$OUT/ablation.md; : > $R
ON="on-w4:spx.gpu.unit_windows=true,spx.gpu.waves=4"
ON8="on-w8:spx.gpu.unit_windows=true,spx.gpu.waves=8"
ON2="on-w2:spx.gpu.unit_windows=true,spx.gpu.waves=2"
OFF="off-w4:spx.gpu.unit_windows=false,spx.gpu.waves=4"
timeout 600 python3 tools/abl.py syn-nlpkkt --edge 240 --steps 40 --header $OFF $ON $ON8 $ON2 $OFF $ON 2>$OUT/full.err | tee -a $R
for v in XW_NOX XW_NOADD XW_NOXADD XW_NOSTAGE; do
    SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_$v.so SPX_BENCH_ABLATION=1 timeout 600 python3 tools/abl.py syn-nlpkkt --edge 240 --steps 40 "$v-$ON" 2>$OUT/$v.err | tee -a $R
done
timeout 600 python3 tools/abl.py syn-nlpkkt --edge 120 $OFF $ON $ON8 $ON2 2>$OUT/e120.err | tee -a $R
for w in syn-cant syn-nd24k syn-webbase syn-kkt2f; do
    timeout 600 python3 tools/abl.py $w --edge 100 "off:spx.gpu.unit_windows=false" "on:spx.gpu.unit_windows=true" "auto:" 2>$OUT/$w.err | tee -a $R
done
