#!/bin/bash
# round-3 probe 4: column phases on syn-webbase; bench.py vs tools/abl.py on the bench matrix
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r03e; mkdir -p $OUT; cd $ROOT
S=$OUT/probe4.md
python tools/abl.py syn-webbase --header --steps 300 p1:spx.gpu.col_phases=1 p2:spx.gpu.col_phases=2 p3:spx.gpu.col_phases=3 p4:spx.gpu.col_phases=4 p5:spx.gpu.col_phases=5 p6:spx.gpu.col_phases=6 p8:spx.gpu.col_phases=8 auto: p1w4:spx.gpu.col_phases=1,spx.gpu.waves=4 p4w4:spx.gpu.col_phases=4,spx.gpu.waves=4 p4w8:spx.gpu.col_phases=4,spx.gpu.waves=8 p4r512:spx.gpu.col_phases=4,spx.gpu.rowblock_elems=1024 p4r4k:spx.gpu.col_phases=4,spx.gpu.rowblock_elems=4096 > $S 2>$OUT/err.txt
python tools/abl.py syn-bandrandom --steps 300 p1:spx.gpu.col_phases=1 auto: >> $S 2>>$OUT/err.txt
python -m pytest tests/test_gpu_parity.py -q -x -k "column_phases" > $OUT/pytest_phases.log 2>&1; tail -n 3 $OUT/pytest_phases.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $OUT/bench_general.json 2>$OUT/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-graph > $OUT/bench_general_nograph.json 2>>$OUT/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --symmetric > $OUT/bench_sym.json 2>>$OUT/bench.err
for f in bench_general bench_general_nograph bench_sym; do python - $OUT/$f.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'], d['format']['index_bytes_per_nnz'], d['config']['launch'][:20])
PY
done
cat $S; tail -n 3 $OUT/err.txt $OUT/bench.err
