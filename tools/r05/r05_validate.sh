#!/bin/bash
# round-5 validation of a tree on the GPU box: the whole -m gpu suite, smoke(), the randomised soak with the
# unit-window options in its option space, the default bench line
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05v; mkdir -p $OUT; cd $ROOT
timeout 3000 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $OUT/pytest_gpu.txt
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 > $OUT/smoke.txt
timeout 1500 python3 tools/soak_random.py ${SOAK_FROM:-1000} ${SOAK_TO:-1400} 2>&1 | tail -12 > $OUT/soak_random.txt
timeout 900 python3 bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench.json
cat $OUT/pytest_gpu.txt $OUT/smoke.txt $OUT/soak_random.txt; head -c 400 $OUT/bench.json
