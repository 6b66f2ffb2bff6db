#!/bin/bash
# Function-level profile (gprof) of a host-only spx_mat_tune on the nlpkkt stand-in.
# usage: tools/prof_tune.sh [edge=60] [threads=1] [symmetric=false]
set -e
cd "$(dirname "$0")/.."
make -s lib
O=build/prof; mkdir -p $O
CS=sparsex_amd/csrc
for f in common config partition stats encoder input reorder csx_emit gpu_emit stream_index dist api; do
    [ $O/$f.o -nt $CS/$f.cpp ] && [ -z "$(find $CS include -name "*.h*" -newer $O/$f.o)" ] || g++ -std=c++17 -O2 -g -fno-omit-frame-pointer -Iinclude -I$CS -pthread -c $CS/$f.cpp -o $O/$f.o
done
gcc -O2 -g -pg -Iinclude -c tools/prof_tune.c -o $O/prof_tune.o
gcc -O3 -c tools/synth/nlpkkt_gen.c -o $O/nlpkkt_gen.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -pg -o $O/prof_tune $O/*.o build/obj/spmv_kernels.o build/obj/vec_kernels.o \
    build/obj/dist_kernels.o -pthread -ldl -lm
(cd $O && rm -f gmon.out && ./prof_tune "${1:-60}" "${2:-1}" "${3:-false}" && gprof -b -p ./prof_tune gmon.out | cut -c1-240 | head -45)
