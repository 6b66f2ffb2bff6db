#!/bin/bash
# round-4 probe 6: the whole GPU suite on the current tree (with a usable trace should anything crash), the per-row
# modes of the column slices A/B on webbase, a Matrix Market file of edge 120 through bench.py --mtx
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r04f; mkdir -p $OUT; cd $ROOT
export PYTHONFAULTHANDLER=1
timeout 1500 python3 -X faulthandler -m pytest tests -x -q -m gpu -k "not multirank" -p no:cacheprovider > $OUT/pytest_gpu_full.txt 2>&1
tail -5 $OUT/pytest_gpu_full.txt
grep -n "Fatal\|Segmentation\|File \"/root/repo\|File \"/tmp" $OUT/pytest_gpu_full.txt | head -30 > $OUT/pytest_crash.txt
R=$OUT/webbase_row_modes_raw.md; : > $R
for i in 1 2; do
python3 tools/abl.py syn-webbase --steps 400 --header "row modes:" "row modes:" 2>/dev/null | tee -a $R
SPX_NO_ROW_MODES=1 python3 tools/abl.py syn-webbase --steps 400 "all rows added (SPX_NO_ROW_MODES):" "all rows added (SPX_NO_ROW_MODES):" 2>/dev/null | tee -a $R
done
python3 tools/mm_write.py syn-nlpkkt /tmp/nlpkkt120.mtx --edge 120 | tee $OUT/mtx.txt
python3 bench.py --mtx /tmp/nlpkkt120.mtx --no-cpu-baseline --no-configs --steps 40 --warmup 5 2> $OUT/bench_mtx.err | tail -1 > $OUT/bench_mtx_general.json
python3 bench.py --mtx /tmp/nlpkkt120.mtx --symmetric --no-cpu-baseline --no-configs --steps 40 --warmup 5 2>> $OUT/bench_mtx.err | tail -1 > $OUT/bench_mtx_symmetric.json
rm -f /tmp/nlpkkt120.mtx
head -c 1500 $OUT/bench_mtx_general.json; echo
