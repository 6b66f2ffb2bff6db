#!/bin/bash
# round 6, VERDICT r05 item 1(a): the write side of the symmetric read-once step, bisected (e240, symmetric path):
# experiment builds of spmv_kernels.hip (spx_abl.hpp) that leave out the own rows / the hand-over of the slots / the
# init pass, alone and together; own rows added instead of stored; and the same on top of the stream-only build.
# Built beforehand with tools/r06/r06_build_variants.sh (the variants travel with the snapshot).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06w; mkdir -p $OUT; cd $ROOT
R=$OUT/sym_writes.md; : > $R
run() { timeout 900 python3 tools/abl.py syn-nlpkkt --edge 240 --symmetric --steps 50 "$@"; }
run --header default: 2>$OUT/abl_default.err | tee -a $R
for v in ${VARIANTS:-SYM_NOOWN SYM_NOHANDOVER SYM_NOINIT SYM_NOWRITES SYM_NOPRIVATE SYM_STREAM SYM_STREAM_NOWRITES}; do
  echo "$v" >> $R
  SPX_BENCH_ABLATION=1 SPX_LIB_PATH=$ROOT/sparsex_amd/lib/variants/libsparsex_$v.so run default: 2>$OUT/abl_$v.err | tee -a $R
done
run default: 2>>$OUT/abl_default.err | tee -a $R
