#!/bin/bash
# builds the experiment variants the round-6 scripts load (cross-compiled here, they travel to the GPU box):
#   tools/r06/r06_build_variants.sh [variant ...]     (default: the symmetric ablations of spx_abl.hpp)
cd "$(dirname "$0")/../.."
export SPX_VARIANT_TU="spmv_kernels spmv_sx_kernels"
V=${@:-SYM_NOOWN SYM_NOHANDOVER SYM_NOINIT SYM_NOWRITES SYM_NOPRIVATE SYM_STREAM SYM_NOX SYM_NOSLOTADD}
make lib > /dev/null
for v in $V; do
  tools/build_variant.sh $v "-DSPX_ABL_$v" &
done
tools/build_variant.sh SYM_STREAM_NOWRITES "-DSPX_ABL_SYM_STREAM -DSPX_ABL_SYM_NOWRITES" &
wait
