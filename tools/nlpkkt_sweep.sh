for e in 2048 4096 8192; do for w in 4 8; do python bench.py --no-cpu-baseline --workload syn-nlpkkt --edge ${EDGE:-120} --steps 200 --warmup 20 --opt spx.gpu.rowblock_elems=$e --opt spx.gpu.waves=$w 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('nlpkkt rbe $e W$w %8.1f GF/s %8.4f ms  frac %.4f  rb %6d idxB %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['format']['rowblocks'], d['format']['index_bytes_per_nnz']))"; done; done
