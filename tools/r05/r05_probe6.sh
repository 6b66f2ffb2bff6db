#!/bin/bash
# round-5 probe 6: is the unit-window kernel bound by its instruction work?  Builds whose second half only
# consumes the loaded values (no decode, no LDS reads, no FMAs, no adds), with and without the staging.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05f; mkdir -p $OUT; cd $ROOT
R=$OUT/abl.md; : > $R
ON="on-w4:spx.gpu.unit_windows=true,spx.gpu.waves=4,spx.gpu.unit_window_doubles=3072"
ON8="on-w8-32k:spx.gpu.unit_windows=true,spx.gpu.waves=8,spx.gpu.rowblock_rows=2048,spx.gpu.rowblock_elems=32768,spx.gpu.unit_window_doubles=12000"
OFF="off-w4:spx.gpu.unit_windows=false,spx.gpu.waves=4"
timeout 600 python3 tools/abl.py syn-nlpkkt --edge 240 --steps 40 --header $OFF $ON $ON8 2>$OUT/full.err | tee -a $R
for v in XW_NOFINISH XW_NOFINISH_NOSTAGE; do
    SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_$v.so SPX_BENCH_ABLATION=1 timeout 600 python3 tools/abl.py syn-nlpkkt --edge 240 --steps 40 "$v-$ON" "$v-$ON8" 2>$OUT/$v.err | tee -a $R
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/stream_pattern.hip -o $OUT/stream_pattern 2>/dev/null
timeout 300 $OUT/stream_pattern 64 30 2>&1 | tail -7 | tee $OUT/pattern.txt
