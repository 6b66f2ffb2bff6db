#!/bin/bash
# read-once row segments (spx.gpu.sym_segments) on and off: syn-nlpkkt at two sizes, syn-cant, syn-nd24k
mkdir -p gpurun_out
out=gpurun_out/symseg_probe.txt
: > $out
line() {
  python bench.py --no-cpu-baseline --no-configs --steps ${STEPS:-100} "$@" 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('%-14s sym=%d %8.1f GF/s %8.4f ms frac %.4f rb %6d W%d idxB/nnz %.3f stored %d kernel %s' % (d['config']['workload'][:14], d['config']['symmetric_path'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['format']['rowblocks'], d['format']['waves_per_workgroup'], d['format']['index_bytes_per_nnz'], d['format'].get('nnz_stored', -1), d['roofline'].get('kernel', '?')))"
}
for e in ${EDGES:-120 190}; do
  echo "== nlpkkt edge $e general" >> $out;   line --edge $e >> $out 2>&1
  for m in true false; do
    echo "== nlpkkt edge $e symmetric segments=$m" >> $out
    line --edge $e --symmetric --opt spx.gpu.sym_segments=$m >> $out 2>&1
  done
done
for w in syn-cant syn-nd24k; do
  for m in true false auto; do
    echo "== $w symmetric segments=$m" >> $out
    STEPS=400 line --workload $w --symmetric --opt spx.gpu.sym_segments=$m >> $out 2>&1
  done
done
cat $out
