#!/bin/bash
# round-5 probe 2: (1) the GPU parity tests of the unit-window kernel; (2) what a read-only stream reaches
# when it is read in the SpMV's access pattern (tools/micro/stream_pattern.hip); (3) ablation builds of the
# unit-window kernel on the bench matrix (x reads / LDS adds / staging compiled out).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05b; mkdir -p $OUT; cd $ROOT
timeout 1200 python3 -m pytest tests/test_gpu_unit_windows.py -q 2>&1 | tail -25 > $OUT/pytest.txt
cat $OUT/pytest.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/stream_pattern.hip -o $OUT/stream_pattern 2>/dev/null
for lds in 0 24 40; do timeout 300 $OUT/stream_pattern 64 $lds; done 2>&1 | tee $OUT/pattern.txt
R=$OUT/ablation.md; : > $R
ON="on-w4:spx.gpu.unit_windows=true,spx.gpu.waves=4"
OFF="off-w4:spx.gpu.unit_windows=false,spx.gpu.waves=4"
timeout 600 python3 tools/abl.py syn-nlpkkt --edge 240 --steps 40 --header $OFF $ON $OFF $ON 2>$OUT/full.err | tee -a $R
for v in XW_NOX XW_NOADD XW_NOXADD XW_NOSTAGE; do
    SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_$v.so SPX_BENCH_ABLATION=1 timeout 600 python3 tools/abl.py syn-nlpkkt --edge 240 --steps 40 "$v-$ON" 2>$OUT/$v.err | tee -a $R
done
