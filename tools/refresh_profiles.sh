#!/bin/bash
# Runs, on the GPU box, the whole set of measurements profiles/<round>/ holds:
# rocprofv3 kernel stats + FETCH_SIZE/WRITE_SIZE passes for the three synthetic
# workloads, then the plain bench lines (with the CPU baseline).
# usage: tools/refresh_profiles.sh <round-tag>       e.g. r01
set -u
R=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for W in cant nd24k webbase; do
    bash "$ROOT/tools/profile.sh" ${R}_$W --workload syn-$W > /dev/null 2>&1
done
cd "$ROOT"
for W in cant nd24k webbase; do
    python3 bench.py --workload syn-$W 2> gpurun_out/${R}_$W/bench_plain.err | tail -1 > gpurun_out/${R}_$W/bench_plain.json
done
python3 bench.py --workload syn-nd24k --symmetric 2>/dev/null | tail -1 > gpurun_out/${R}_nd24k/bench_plain_sym.json
python3 bench.py --workload syn-cant --symmetric 2>/dev/null | tail -1 > gpurun_out/${R}_cant/bench_plain_sym.json
for W in cant nd24k webbase; do
    echo "== $W"; cat gpurun_out/${R}_$W/bench_plain.json; head -4 gpurun_out/${R}_$W/kernel_stats.csv; cat gpurun_out/${R}_$W/pmc_*.txt | grep csx_spmv
done
