#!/bin/bash
# round 6: validation of the tree -- the whole `-m gpu` suite, smoke(), the default bench line
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06v; mkdir -p $OUT; cd $ROOT
timeout 3000 python3 -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
timeout 1200 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1500 $OUT/bench_default.json | head -c 1500; echo
