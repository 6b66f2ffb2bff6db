#!/bin/bash
# round 6: spx_matvec_mult on page-locked host vectors of the bench matrix, number of parts (spx.rt.host_parts) swept
# inside one process (boxes differ by more than the steps do)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06h; mkdir -p $OUT; cd $ROOT
timeout 1200 python3 - > $OUT/host_parts_sweep.txt 2>&1 <<'PY'
import ctypes as C, time, numpy as np, sys, os
sys.path.insert(0, "tests")
import sparsex_amd as sx
from sparsex_amd import synth
from sparsex_amd.api import VectorStruct
from helpers import tune
L = sx.lib()
csr = synth.syn_nlpkkt(240)
n = csr[3]
L.spx_vec_create_random.restype = C.POINTER(VectorStruct); L.spx_vec_create_random.argtypes = [C.c_size_t, C.c_void_p]
L.spx_vec_create.restype = C.POINTER(VectorStruct); L.spx_vec_create.argtypes = [C.c_size_t, C.c_void_p]
L.spx_mat_get_partition.restype = C.c_void_p
for sym in (False, True):
    A = tune(csr, {"spx.matrix.symmetric": "true"} if sym else {}, sym=sym)
    part = C.c_void_p(L.spx_mat_get_partition(C.c_void_p(A.handle)))
    xv, yv = L.spx_vec_create_random(n, part), L.spx_vec_create(n, part)
    for rnd in range(2):
        for parts in (8, 16, 24, 32, 48, 64, 0):
            sx.option_set("spx.rt.host_parts", str(parts))
            for rep in range(2):
                L.spx_matvec_mult(C.c_double(0.5), C.c_void_p(A.handle), xv, yv)
            t0 = time.perf_counter()
            for rep in range(10):
                L.spx_matvec_mult(C.c_double(0.5), C.c_void_p(A.handle), xv, yv)
            print("symmetric" if sym else "general", "parts", parts, "%.2f ms per spx_matvec_mult" % ((time.perf_counter() - t0) * 100), flush=True)
    L.spx_vec_destroy(xv); L.spx_vec_destroy(yv)
    A.destroy()
PY
cat $OUT/host_parts_sweep.txt | grep parts
