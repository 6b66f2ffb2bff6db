#!/bin/bash
# round 4: the committed evidence of the final tree -- rocprofv3 kernel stats + PMC passes + plain bench lines
# (tools/refresh_profiles.sh), then the randomised parity soak
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
bash tools/refresh_profiles.sh r04 > gpurun_out/r04_refresh.log 2>&1
tail -30 gpurun_out/r04_refresh.log | cut -c1-220
bash tools/r04/r04_soak.sh 2>&1 | tail -20
