#!/bin/bash
# round-4 probe 8: the read-once segment kernel forced to 8 wavefronts per SIMD (64 VGPRs + 20 B of scratch instead of
# 67 VGPRs / 7 wavefronts), alternating processes; then the multi-rank tests on the final tree
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r04h; mkdir -p $OUT; cd $ROOT
bash tools/build_variant.sh w8 "-DSPX_SYMSEG_NOTILE_ATTR=__attribute__((amdgpu_waves_per_eu(8,8)))" > $OUT/build.txt 2>&1
V=$ROOT/sparsex_amd/lib/variants/libsparsex_w8.so
R=$OUT/symseg_waves8_raw.md; : > $R
for i in 1 2; do
  echo "product build (67 VGPRs, 7 wavefronts / SIMD)" >> $R
  python3 tools/abl.py syn-nlpkkt --edge 240 --symmetric --steps 60 "default:" 2>/dev/null | tee -a $R
  echo "amdgpu_waves_per_eu(8, 8): 64 VGPRs + 20 B scratch" >> $R
  SPX_LIB_PATH=$V python3 tools/abl.py syn-nlpkkt --edge 240 --symmetric --steps 60 "waves_per_eu 8:" 2>/dev/null | tee -a $R
done
python3 tools/abl.py syn-kkt2f --edge 100 --symmetric "default:" 2>/dev/null | tee -a $R
SPX_LIB_PATH=$V python3 tools/abl.py syn-kkt2f --edge 100 --symmetric "waves_per_eu 8:" 2>/dev/null | tee -a $R
python3 tools/abl.py syn-nlpkkt --edge 120 --symmetric "default:" 2>/dev/null | tee -a $R
SPX_LIB_PATH=$V python3 tools/abl.py syn-nlpkkt --edge 120 --symmetric "waves_per_eu 8:" 2>/dev/null | tee -a $R
timeout 2400 python3 -m pytest tests/test_gpu_multirank.py -x -q -m gpu --durations=5 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -15 | tee $OUT/pytest_multirank.txt
