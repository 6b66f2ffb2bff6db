#!/bin/bash
# Runs, on the GPU box, the whole set of measurements profiles/<round>/ holds:
# rocprofv3 kernel stats + FETCH_SIZE/WRITE_SIZE passes (separate runs) of the
# default bench command (syn-nlpkkt, general path; and its symmetric path) and of the three BASELINE
# configurations of the "configs" object, then the plain bench lines (with the
# CPU baselines).
# usage: tools/refresh_profiles.sh <round-tag>       e.g. r02
set -u
R=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
bash "$ROOT/tools/profile.sh" ${R}_nlpkkt > /dev/null 2>&1
bash "$ROOT/tools/profile.sh" ${R}_nlpkkt_sym --symmetric > /dev/null 2>&1
bash "$ROOT/tools/profile.sh" ${R}_cant --workload syn-cant > /dev/null 2>&1
bash "$ROOT/tools/profile.sh" ${R}_nd24k_sym --workload syn-nd24k --symmetric > /dev/null 2>&1
bash "$ROOT/tools/profile.sh" ${R}_webbase --workload syn-webbase > /dev/null 2>&1
cd "$ROOT"
python3 bench.py 2> gpurun_out/${R}_nlpkkt/bench_plain.err | tail -1 > gpurun_out/${R}_nlpkkt/bench_plain.json
python3 bench.py --symmetric --no-configs 2>/dev/null | tail -1 > gpurun_out/${R}_nlpkkt/bench_plain_sym.json
python3 bench.py --workload syn-nd24k --no-configs 2>/dev/null | tail -1 > gpurun_out/${R}_nd24k_sym/bench_plain_general.json
for W in nlpkkt nlpkkt_sym cant nd24k_sym webbase; do
    echo "== $W"; head -c 600 gpurun_out/${R}_$W/bench_line.json; echo; head -6 gpurun_out/${R}_$W/kernel_stats.csv | cut -c1-200; grep csx_ gpurun_out/${R}_$W/pmc_*.txt | cut -c1-200
done
