#!/bin/bash
# round 6: the host-vector entry point on symmetric streams whose parts can run in any order (KKT systems of several
# sizes and option sets, the thresholds lowered): library vectors and views kept across calls, spx_matvec_mult and
# spx_matvec_kernel (beta != 0: y by need), every result against CSR; a product with a NaN x first
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06s; mkdir -p $OUT; cd $ROOT
export SPX_HOST_PARTS_MIN_BYTES=1024
for XP in 4096 32768; do
SPX_HOST_XPIECE_BYTES=$XP timeout 1500 python3 - >> $OUT/soak_host_sym.txt 2>&1 <<'PY'
import ctypes as C, itertools, os, sys, numpy as np
sys.path.insert(0, "tests")
import sparsex_amd as sx
from sparsex_amd import synth
from sparsex_amd.api import VectorStruct
from helpers import tune, check_y
L = sx.lib()
if os.environ.get("R06_LOG") == "1":
    L.spx_log_info_console()
L.spx_vec_create_random.restype = C.POINTER(VectorStruct); L.spx_vec_create_random.argtypes = [C.c_size_t, C.c_void_p]
L.spx_vec_create.restype = C.POINTER(VectorStruct); L.spx_vec_create.argtypes = [C.c_size_t, C.c_void_p]
L.spx_vec_create_from_buff.restype = C.POINTER(VectorStruct)
L.spx_vec_create_from_buff.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
L.spx_mat_get_partition.restype = C.c_void_p
L.spx_matvec_kernel.argtypes = [C.c_double, C.c_void_p, C.POINTER(VectorStruct), C.c_double, C.POINTER(VectorStruct)]
OPTS = [{}, {"spx.gpu.sym_pipeline": "false"}, {"spx.gpu.sym_wide_rows": "512"}, {"spx.gpu.waves": "4"},
        {"spx.gpu.sym_segment_max": "4"}, {"spx.rt.host_parts": "7"}, {"spx.rt.host_parts": "40"}]
bad = ran = by_need = 0
for edge, (k, extra) in itertools.product((36, 40, 44, 48, 52, 56), enumerate(OPTS)):
    if (edge // 4 + k) % 2:          # (half of the grid)
        continue
    for sym in (True, False):
        o = dict(extra)
        if os.environ.get("R06_NO_REGISTER") == "1":
            o["spx.vec.register"] = "false"
        if sym:
            o.update({"spx.gpu.sym_segments": "true"})
        csr = synth.syn_nlpkkt(edge)
        n = csr[3]
        try:
            A = tune(csr, o, sym=sym)
            part = C.c_void_p(L.spx_mat_get_partition(C.c_void_p(A.handle)))
            x, y0 = synth.random_x(n, seed=edge + k), synth.random_x(n, seed=edge + k + 1)
            for kind in ("library", "views"):
                if kind == "library":
                    xv, yv = L.spx_vec_create_random(n, part), L.spx_vec_create(n, part)
                    xa = np.ctypeslib.as_array(xv.contents.elements, shape=(n,))
                    ya = np.ctypeslib.as_array(yv.contents.elements, shape=(n,))
                else:
                    xa, ya = np.empty(n), np.empty(n)
                    xv = L.spx_vec_create_from_buff(xa.ctypes.data, None, n, None, 43)
                    yv = L.spx_vec_create_from_buff(ya.ctypes.data, None, n, None, 43)
                xa[:] = np.nan
                ya[:] = 0.0
                assert L.spx_matvec_mult(C.c_double(0.5), C.c_void_p(A.handle), xv, yv) == 0
                xa[:] = x
                ya[:] = np.nan
                assert L.spx_matvec_mult(C.c_double(0.5), C.c_void_p(A.handle), xv, yv) == 0
                check_y(csr, x, ya.copy(), 0.5)
                by_need += int(len(A.host_order()) >= 2)
                ya[:] = y0
                assert L.spx_matvec_kernel(2.0, C.c_void_p(A.handle), xv, -0.5, yv) == 0
                check_y(csr, x, ya.copy(), 2.0, -0.5, y0)
                L.spx_vec_destroy(xv); L.spx_vec_destroy(yv)
                ran += 1
                print("ok edge %d sym %s %s %s parts %d order %s" % (edge, sym, o, kind, A.host_parts(), A.host_order()), flush=True)
            A.destroy()
        except Exception as e:
            bad += 1
            print("FAILED edge %d sym %s %s: %s %s" % (edge, sym, o, type(e).__name__, str(e)[:200]), flush=True)
        sx.options_reset()
print("x pieces of %s bytes: %d products checked, x by need in %d, %d failures" % (os.environ["SPX_HOST_XPIECE_BYTES"], ran, by_need, bad))
PY
done
grep -E 'products checked|FAILED' $OUT/soak_host_sym.txt | tail -6
