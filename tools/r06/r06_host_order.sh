#!/bin/bash
# round 6: spx_matvec_mult on host vectors, x sent in the order the parts of the product need it: the order the
# plan chooses on the bench matrix (log line; R06_SYM=1: its symmetric path) and the time per call on library vectors
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06h; mkdir -p $OUT; cd $ROOT
timeout 900 python3 - > $OUT/host_order.txt 2>&1 <<'PY'
import ctypes as C, time, numpy as np, sys
sys.path.insert(0, "tests")
import sparsex_amd as sx
from sparsex_amd import synth
from sparsex_amd.api import VectorStruct
from helpers import tune
L = sx.lib()
L.spx_log_info_console()
csr = synth.syn_nlpkkt(240)
n = csr[3]
import os
SYM = os.environ.get("R06_SYM") == "1"
A = tune(csr, {"spx.matrix.symmetric": "true"} if SYM else {}, sym=SYM)
L.spx_vec_create_random.restype = C.POINTER(VectorStruct); L.spx_vec_create_random.argtypes = [C.c_size_t, C.c_void_p]
L.spx_vec_create.restype = C.POINTER(VectorStruct); L.spx_vec_create.argtypes = [C.c_size_t, C.c_void_p]
L.spx_mat_get_partition.restype = C.c_void_p
part = C.c_void_p(L.spx_mat_get_partition(C.c_void_p(A.handle)))
xv, yv = L.spx_vec_create_random(n, part), L.spx_vec_create(n, part)
for rep in range(3):
    L.spx_matvec_mult(C.c_double(0.5), C.c_void_p(A.handle), xv, yv)
t0 = time.perf_counter()
for rep in range(10):
    L.spx_matvec_mult(C.c_double(0.5), C.c_void_p(A.handle), xv, yv)
print("library vectors: %.2f ms per spx_matvec_mult" % ((time.perf_counter() - t0) * 100))
L.spx_matvec_kernel.argtypes = [C.c_double, C.c_void_p, C.POINTER(VectorStruct), C.c_double, C.POINTER(VectorStruct)]
for rep in range(2):
    L.spx_matvec_kernel(0.5, C.c_void_p(A.handle), xv, 0.25, yv)
t0 = time.perf_counter()
for rep in range(10):
    L.spx_matvec_kernel(0.5, C.c_void_p(A.handle), xv, 0.25, yv)
print("library vectors: %.2f ms per spx_matvec_kernel (beta != 0: y travels both ways)" % ((time.perf_counter() - t0) * 100))
PY
grep -E 'host vectors|library vectors' $OUT/host_order.txt | cut -c1-400
