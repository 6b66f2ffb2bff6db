#!/bin/bash
# Function-level profile (gprof) of a host-only spx_mat_tune on the nlpkkt stand-in.
# usage: tools/prof_tune.sh [edge=60] [threads=1] [symmetric=false] [reorder]
set -e
cd "$(dirname "$0")/.."
make -s lib
O=build/prof; mkdir -p $O
CS=sparsex_amd/csrc
# (the host sources and the HIP objects are the Makefile's lists)
for src in $(make -s print-host-srcs); do
    f=${src%.cpp}
    [ $O/$f.o -nt $CS/$f.cpp ] && [ -z "$(find $CS include -name "*.h*" -newer $O/$f.o)" ] || g++ -std=c++17 -O2 -g -fno-omit-frame-pointer -Iinclude -I$CS -pthread -c $CS/$f.cpp -o $O/$f.o
done
gcc -O2 -g -pg -Iinclude -c tools/prof_tune.c -o $O/prof_tune.o
gcc -O3 -c tools/synth/nlpkkt_gen.c -o $O/nlpkkt_gen.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -pg -o $O/prof_tune $O/*.o $(make -s print-hip-objs) -pthread -ldl -lm
(cd $O && rm -f gmon.out && ./prof_tune "${1:-60}" "${2:-1}" "${3:-false}" "${4:-}" && gprof -b -p ./prof_tune gmon.out | cut -c1-240 | head -45)
