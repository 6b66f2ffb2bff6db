#!/bin/bash
# round 3: everything profiles/r03/ holds (rocprofv3 kernel stats, PMC passes, plain bench lines), then where
# the read-once segment kernel's time goes on the bench matrix (ablation builds, made on the box)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
bash tools/refresh_profiles.sh r03 > gpurun_out/r03_refresh.log 2>&1
tail -n 40 gpurun_out/r03_refresh.log
for v in SEG_NOFLUSH SEG_NOSLOTADD; do bash tools/build_variant.sh $v "-DSPX_ABL_$v" > /dev/null 2>&1; done
bash tools/build_variant.sh SEG_NEITHER "-DSPX_ABL_SEG_NOFLUSH -DSPX_ABL_SEG_NOSLOTADD" > /dev/null 2>&1
EDGE=240 bash tools/ablate_seg.sh > gpurun_out/r03_symseg_ablation_e240.txt 2>&1
cat gpurun_out/r03_symseg_ablation_e240.txt
