#!/bin/bash
# round-4 probe 15: the bench line of the contract matrix on eight ranks sharing the one GPU (gloo), renumbered (the default)
# and in the application's order: the `collective` object and the per-rank facts as committed evidence
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r04o; mkdir -p $OUT; cd $ROOT
export SPX_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
for MODE in rcm_owner none; do
  PORT=$(python3 -c "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])")
  timeout 1500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus 8 --steps 3 --warmup 1 \
     --no-cpu-baseline --no-configs --host-threads 2 --dist-reorder $MODE 2> $OUT/err_$MODE.log | grep '^{"metric"' | tail -1 > $OUT/bench_eight_ranks_one_gpu_$MODE.json
  python3 -c "
import json
d = json.load(open('$OUT/bench_eight_ranks_one_gpu_$MODE.json')); c = d['collective']
print('$MODE', 'halo MB per rank', [round(8e-6 * r['halo_entries_received'], 2) for r in d['ranks']], 'parts', [r['overlap_parts'] for r in d['ranks']], 'rounds', c['overlap_rounds'], 'reorder s', d['config']['dist_reorder_seconds'], 'parity', d['parity']['max_err_over_fp64_bound'])"
done
