#!/bin/bash
# round-4 probe 11: multi-rank tests on the restructured overlapped step, and a kernel trace of one rank of a two-rank
# run sharing the GPU (the ranks are started by hand, one rocprofv3 each: no launcher between the profiler and python)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r04k; mkdir -p $OUT; cd $ROOT
if [ "${SKIP_TESTS:-0}" != "1" ]; then timeout 2400 python3 -m pytest tests/test_gpu_multirank.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 | tee $OUT/pytest_multirank.txt; fi
export SPX_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1 MASTER_PORT=$(python3 -c "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])") WORLD_SIZE=2 HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
ARGS="--gpus 2 --edge 120 --steps 5 --warmup 2 --no-cpu-baseline --no-configs --host-threads 8 --dist-reorder none"
RANK=1 LOCAL_RANK=1 python3 $ROOT/bench.py $ARGS > $OUT/rank1.log 2>&1 &
P1=$!
RANK=0 LOCAL_RANK=0 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace0 -o run -- python3 $ROOT/bench.py $ARGS > $OUT/rank0.log 2>&1
wait $P1
cd $ROOT
grep '^{"metric"' $OUT/rank0.log | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d['collective']
print('full step %.3f ms, not overlapped %.3f ms, gather-y step %.3f ms, kernels only %.3f ms, rounds %d, halo %d B' % (c['full_step_ms'], c['halo_step_not_overlapped_ms'], c['gather_y_step_ms'], c['kernels_only_ms'], c['overlap_rounds'], c['halo_bytes_received_per_rank']))" | tee $OUT/step.txt
python3 tools/overlap_trace.py $OUT/trace0 3 | tee $OUT/overlap_trace.md
find $OUT/trace0 -name '*kernel_trace.csv' -exec cp {} $OUT/rank0_kernel_trace.csv \;
rm -rf $OUT/trace0
