#!/bin/bash
# prints one line per workload: GFLOP/s, ms/step, roofline fraction, row-blocks, index B/nnz
for w in ${WORKLOADS:-syn-cant syn-nd24k syn-webbase}; do
  python bench.py --no-cpu-baseline --no-configs --steps ${STEPS:-400} --workload $w "$@" 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('%-12s %8.1f GF/s %8.4f ms  frac %.4f  rb %6d W%d idxB/nnz %.3f units %d' % (d['config']['workload'][:11], d['value'], d['ms_per_step'], d['roofline']['frac'], d['format']['rowblocks'], d['format']['waves_per_workgroup'], d['format']['index_bytes_per_nnz'], d['format']['units']))"
done
