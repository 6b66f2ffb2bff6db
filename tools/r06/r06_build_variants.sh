#!/bin/bash
# builds the experiment variants the round-6 scripts load (cross-compiled here, they travel to the GPU box)
cd "$(dirname "$0")/../.."
export SPX_VARIANT_TU=spmv_kernels
for v in SYM_NOOWN SYM_NOHANDOVER SYM_NOINIT SYM_NOWRITES SYM_NOPRIVATE SYM_STREAM; do
  tools/build_variant.sh $v "-DSPX_ABL_$v" &
done
tools/build_variant.sh SYM_STREAM_NOWRITES "-DSPX_ABL_SYM_STREAM -DSPX_ABL_SYM_NOWRITES" &
wait
