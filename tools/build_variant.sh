#!/bin/bash
# Builds an experiment variant of the library: the HIP kernels recompiled with
# extra flags, linked with the regular host objects.
# usage: tools/build_variant.sh <name> "<extra hipcc flags>"   ->  sparsex_amd/lib/variants/libsparsex_<name>.so
#        then run with SPX_LIB_PATH=sparsex_amd/lib/variants/libsparsex_<name>.so
set -e
NAME=$1; FLAGS=$2
cd "$(dirname "$0")/.."
make lib > /dev/null
mkdir -p build/var sparsex_amd/lib/variants
# (SPX_VARIANT_TU="spmv_xw_kernels" rebuilds only that translation unit: seconds instead of minutes)
VAR_OBJS=""
for tu in ${SPX_VARIANT_TU:-spmv_kernels spmv_xw_kernels spmv_sx_kernels}; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -munsafe-fp-atomics -Iinclude -Isparsex_amd/csrc \
        $FLAGS -c sparsex_amd/csrc/$tu.hip -o build/var/${tu}_$NAME.o
    VAR_OBJS="$VAR_OBJS build/var/${tu}_$NAME.o"
done
OBJS=$(ls build/obj/*.o)
for tu in ${SPX_VARIANT_TU:-spmv_kernels spmv_xw_kernels spmv_sx_kernels}; do OBJS=$(echo "$OBJS" | grep -v "/$tu.o"); done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o sparsex_amd/lib/variants/libsparsex_$NAME.so $OBJS $VAR_OBJS -pthread -ldl
echo sparsex_amd/lib/variants/libsparsex_$NAME.so
