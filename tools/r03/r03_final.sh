#!/bin/bash
# round 3, final: GPU test-suite, then everything profiles/r03/ holds with the final library
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
(time python -m pytest tests -m gpu -q -n 4) > gpurun_out/r03_final_pytest.log 2>&1; tail -n 5 gpurun_out/r03_final_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
bash tools/refresh_profiles.sh r03 > gpurun_out/r03_refresh.log 2>&1
tail -n 12 gpurun_out/r03_refresh.log | cut -c1-300
