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
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -munsafe-fp-atomics -Iinclude -Isparsex_amd/csrc \
    $FLAGS -c sparsex_amd/csrc/spmv_kernels.hip -o build/var/spmv_$NAME.o
OBJS=$(ls build/obj/*.o | grep -v spmv_kernels.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o sparsex_amd/lib/variants/libsparsex_$NAME.so $OBJS build/var/spmv_$NAME.o -pthread -ldl
echo sparsex_amd/lib/variants/libsparsex_$NAME.so
