#!/bin/bash
# read-once segments with row-blocks of 512 / 1024 / 2048 rows (spx.gpu.sym_wide_rows) on syn-nlpkkt
mkdir -p gpurun_out
out=gpurun_out/wide_probe.txt
: > $out
for e in ${EDGES:-80 120 190}; do
  for w in ${WIDES:-512 1024 2048}; do
    STEPS=100; [ $e -ge 190 ] && STEPS=30
    echo "edge $e wide $w" >> $out
    python bench.py --no-cpu-baseline --no-configs --steps $STEPS --edge $e --symmetric --opt spx.gpu.sym_segments=true --opt spx.gpu.sym_wide_rows=$w "$@" 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('  %8.1f GF/s %8.4f ms rb %6d W%d idxB/nnz %.3f emit %.2fs err/bound %.3f' % (d['value'], d['ms_per_step'], d['format']['rowblocks'], d['format']['waves_per_workgroup'], d['format']['index_bytes_per_nnz'], d['format']['emit_upload_seconds'], d['parity']['max_err_over_fp64_bound']))" >> $out 2>&1
  done
done
cat $out
