#!/bin/bash
# rocprofv3 kernel durations (GPU timestamps) of the ablation variants
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
export SPX_BENCH_ABLATION=1
for w in ${WORKLOADS:-syn-cant}; do
  for v in ${VARIANTS:-FULL EMPTY NOPASS VALSONLY NOX}; do
    if [ $v = FULL ]; then unset SPX_LIB_PATH; else export SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_$v.so; fi
    OUT=$ROOT/gpurun_out/ablp_${w}_$v
    rm -rf $OUT; mkdir -p $OUT
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 $ROOT/bench.py --no-cpu-baseline --no-configs --steps 300 --warmup 30 --workload $w > $OUT/log 2>&1
    echo "$w $v $(find $OUT -name '*kernel_stats.csv' -exec grep csx_spmv {} \; | cut -d, -f2-4,6-7)"
    rm -rf $OUT
  done
done
