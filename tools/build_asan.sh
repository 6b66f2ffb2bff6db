#!/bin/bash
# Host-side AddressSanitizer build of the library (CPU container only: the preprocessor, the
# stream emitter, the index code and the C API are instrumented; the HIP objects are the regular
# ones).  Run the CPU tests against it:
#   tools/build_asan.sh && LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so.6)" \
#       ASAN_OPTIONS=detect_leaks=0 SPX_LIB_PATH=$PWD/sparsex_amd/lib/variants/libsparsex_asan.so \
#       python -m pytest tests -q -m "not gpu" -p no:cacheprovider
# (libstdc++ preloaded too, or the sanitizer cannot intercept the library's exceptions inside python; the link
# tests of tests/test_c_abi.py look for libsparsex.so next to the loaded library and fail by construction)
set -e
cd "$(dirname "$0")/.."
make lib > /dev/null
mkdir -p build/asan sparsex_amd/lib/variants
# the host sources and the HIP objects are the Makefile's lists
HOST=$(make -s -f Makefile print-host-srcs)
HIPO=$(make -s -f Makefile print-hip-objs)
rm -f build/asan/*.o
for f in $HOST; do
    g++ -std=c++17 -O1 -g -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer \
        -Iinclude -Isparsex_amd/csrc -pthread -c sparsex_amd/csrc/$f -o build/asan/${f%.cpp}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o sparsex_amd/lib/variants/libsparsex_asan.so \
    build/asan/*.o $HIPO -pthread -ldl -lubsan \
    -L$(dirname $(gcc -print-file-name=libasan.so)) -lasan
echo sparsex_amd/lib/variants/libsparsex_asan.so
