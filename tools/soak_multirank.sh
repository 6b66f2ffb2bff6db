#!/bin/bash
# one-off: bench.py over several ranks sharing the GPU (gloo), varied sizes and options; the bench's parity gates are the check
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export SPX_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
fail=0
run() {
  w=$1; shift
  port=$(python3 -c "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])")
  out=$(timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $port bench.py --gpus $w --steps 3 --warmup 1 --no-cpu-baseline --host-threads 2 "$@" 2>&1 | grep -E '^\{"metric"|Error|error|assert' | head -3)
  if echo "$out" | grep -q '^{"metric"'; then echo "ok   world $w $*  $(echo "$out" | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['parity']['max_err_over_fp64_bound'], d['roofline']['kernel'][:60])")"; else echo "FAIL world $w $*"; echo "$out" | cut -c1-300; fail=1; fi
}
for w in 2 3 4; do
  run $w --edge 30
  # round 4: the halo exchange in rounds behind 2 / 7 parts of the product; the three numberings
  run $w --edge 33 --dist-reorder none --opt spx.rt.dist_chunks=2 --opt spx.gpu.rowblock_elems=600
  run $w --edge 33 --dist-reorder rcm --opt spx.rt.dist_chunks=7 --opt spx.gpu.rowblock_elems=600
  run $w --edge 41 --dist-reorder rcm_owner --opt spx.rt.dist_chunks=1
  run $w --workload syn-webbase --scale 0.2 --opt spx.rt.dist_chunks=3 --opt spx.gpu.rowblock_elems=500
  run $w --workload syn-cant --scale 0.6 --dist-reorder rcm --opt spx.gpu.rowblock_elems=400
  run $w --workload syn-kkt2f --edge 24
  run $w --workload syn-kkt2f --edge 24 --symmetric --opt spx.gpu.sym_segments=true
  for e in 22 37; do
    run $w --edge $e --symmetric --opt spx.gpu.sym_segments=true --opt spx.gpu.sym_wide_rows=2048
    run $w --edge $e --symmetric --opt spx.gpu.sym_segments=true --opt spx.gpu.sym_wide_rows=512 --opt spx.gpu.sym_segment_min=4
    run $w --edge $e --symmetric --opt spx.gpu.sym_spill=lists
    run $w --edge $e --opt spx.gpu.rowblock_rows=2048
  done
  # round 6: slices whose runs are long enough for passes of their own -- the pipelined read-once kernel on a slice
  # (limited init range, conflict rows, the exchange behind it), in the application's order and renumbered
  run $w --edge 56 --symmetric --dist-reorder none --opt spx.gpu.sym_segments=true --opt spx.gpu.sym_pipeline=true
  run $w --edge 50 --symmetric --opt spx.gpu.sym_segments=true --opt spx.gpu.sym_pipeline=true --opt spx.gpu.waves=4
  run $w --edge 56 --symmetric --dist-reorder none --opt spx.gpu.sym_segments=true --opt spx.gpu.sym_pipeline=false
  run $w --workload syn-nd24k --scale 0.2 --symmetric --opt spx.gpu.sym_segments=true
  run $w --workload syn-cant --scale 0.5 --symmetric --opt spx.gpu.sym_segments=true --opt spx.gpu.sym_wide_rows=1024
  run $w --workload syn-cant --scale 0.5 --symmetric --opt spx.gpu.deterministic=true
done
echo "fail=$fail"
