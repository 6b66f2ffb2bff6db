"""spx_matvec_mult / spx_matvec_kernel on HOST vectors (reference src/api/matvec.c:551-620) where y is large: the
product runs in parts of whole rows and every part's rows travel back while the next part runs
(device_spmv_host).  SPX_HOST_PARTS_MIN_BYTES (read once per process) lowers the size from which that happens, so
that small matrices exercise it: both kinds of vectors (views of user buffers, vectors the library created), beta
zero and non-zero, matrices the stream of which can and cannot be cut -- among them (round 6) symmetric streams whose
row-blocks all store their own rows: those rows come back behind their parts, the rest behind the last one.
On general streams x goes up piece by piece in the order the parts need it (SPX_HOST_XPIECE_BYTES makes the pieces
small enough for these matrices to have many) and the parts run in that order: spx_hip_mat_host_order."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes as C, sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import sparsex_amd as sx
from sparsex_amd import synth
from sparsex_amd.api import VectorStruct
from helpers import tune, check_y
cases = [("kkt", synth.syn_nlpkkt(22), {}), ("cant", synth.syn_cant(0.4), {}), ("web", synth.syn_webbase(0.2), {}),
         ("kkt-sym", synth.syn_nlpkkt(16), {"spx.matrix.symmetric": "true"}),
         ("kkt-plain", synth.syn_nlpkkt(22), {"spx.gpu.unit_windows": "false", "spx.gpu.waves": "8"}),
         # symmetric, read-once segments: every row-block stores its own rows (the multiplier rows of the KKT system) --
         # those travel back part by part, the state rows (which receive sums until the last row-block has run) at the end
         ("kkt-sym-segments", synth.syn_nlpkkt(44), {"spx.matrix.symmetric": "true", "spx.gpu.sym_segments": "true"}),
         ("kkt-sym-segments-plain", synth.syn_nlpkkt(40), {"spx.matrix.symmetric": "true", "spx.gpu.sym_segments": "true",
                                                           "spx.gpu.sym_pipeline": "false", "spx.gpu.sym_wide_rows": "512"}),
         # symmetric tiles: rows are added to from everywhere, the stream is not cut
         ("nd24k-sym", synth.syn_nd24k(0.1), {"spx.matrix.symmetric": "true", "spx.gpu.sym_spill": "atomic"})]
L = sx.lib()
L.spx_hip_mat_host_parts.restype = C.c_int
L.spx_hip_mat_host_order.restype = C.c_int
L.spx_hip_vec_page_locked.restype = C.c_int
L.spx_hip_vec_page_locked.argtypes = [C.POINTER(VectorStruct)]
L.spx_hip_mat_host_order.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int]
def host_order(A):
    buf = (C.c_int32 * 32)()
    k = L.spx_hip_mat_host_order(C.c_void_p(A.handle), buf, 32)
    return [buf[i] for i in range(min(k, 32))]
L.spx_vec_create_random.restype = C.POINTER(VectorStruct); L.spx_vec_create_random.argtypes = [C.c_size_t, C.c_void_p]
L.spx_vec_create.restype = C.POINTER(VectorStruct); L.spx_vec_create.argtypes = [C.c_size_t, C.c_void_p]
L.spx_mat_get_partition.restype = C.c_void_p
orders = {}
for name, csr, opts in cases:
    n = csr[3]
    A = tune(csr, opts, sym=opts.get("spx.matrix.symmetric") == "true")
    x = synth.random_x(n)
    # views of user buffers (first a product with another x: a piece of x that the plan forgot would still hold it)
    y = np.full(n, np.nan)
    A.matvec_mult(0.5, np.full(n, np.nan), y)
    A.matvec_mult(0.5, x, y)
    check_y(csr, x, y, 0.5)
    parts = L.spx_hip_mat_host_parts(C.c_void_p(A.handle))
    if not %(expect_parts)s or name in ("kkt-sym", "nd24k-sym"):    # (symmetric streams whose rows are added to from elsewhere are not cut)
        assert parts == 0, (name, parts)
    elif name != "web":                                             # (web: cut unless the tuner chose column slices)
        assert parts >= 2, (name, parts)
    o = host_order(A)        # (a view's buffer is page-locked in place at its first product: by need then; else staged)
    assert o == [] or sorted(o) == list(range(parts)), (name, parts, o)
    y0 = synth.random_x(n, seed=3)
    y = y0.copy()
    A.matvec_kernel(2.0, x, -0.5, y)
    check_y(csr, x, y, 2.0, -0.5, y0)
    # a view that lives across products (the reference's bench loop, src/bench/SparsexModule.cpp:54-70).  A buffer of
    # 32 MB or more is page-locked in place at the first product (spx.vec.register) and travels without staging --
    # tests/test_gpu_fullsize.py goes that way; these small ones are staged (spx_hip_vec_page_locked == 0)
    L.spx_vec_create_from_buff.restype = C.POINTER(VectorStruct)
    L.spx_vec_create_from_buff.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
    xb, yb = synth.random_x(n, seed=5), np.full(n, np.nan)
    xw = L.spx_vec_create_from_buff(xb.ctypes.data, None, n, None, 43)       # SPX_VEC_AS_IS
    yw = L.spx_vec_create_from_buff(yb.ctypes.data, None, n, None, 43)
    for rep in range(4):
        if rep == 2:
            xb[:] = synth.random_x(n, seed=6)          # (the client writes through its own pointer)
        yb[:] = np.nan
        assert L.spx_matvec_mult(C.c_double(0.5), C.c_void_p(A.handle), xw, yw) == 0
        check_y(csr, xb.copy(), yb.copy(), 0.5)
        order, parts_now = host_order(A), L.spx_hip_mat_host_parts(C.c_void_p(A.handle))
        locked = L.spx_hip_vec_page_locked(xw)
        assert locked in (0, 2, 3), (name, rep, locked)
        if parts >= 2 and locked:      # (a box that cannot page-lock client memory stages it: locked == 0)
            # page-locked: x went up by need, every part ran once, in the order the plan chose (symmetric
            # streams too: the init pass goes in front of whichever part runs first)
            assert sorted(order) == list(range(parts_now)) and parts_now >= 2, (name, rep, parts_now, order)
            orders[name] = order
        else:
            assert order == [], (name, rep, order)
    yb[:] = y0
    L.spx_matvec_kernel.argtypes = [C.c_double, C.c_void_p, C.POINTER(VectorStruct), C.c_double, C.POINTER(VectorStruct)]
    assert L.spx_matvec_kernel(2.0, C.c_void_p(A.handle), xw, -0.5, yw) == 0
    check_y(csr, xb.copy(), yb.copy(), 2.0, -0.5, y0)
    L.spx_vec_destroy(xw); L.spx_vec_destroy(yw)
    # vectors the library created (page-locked, x resident between calls)
    part = C.c_void_p(L.spx_mat_get_partition(C.c_void_p(A.handle)))
    xv, yv = L.spx_vec_create_random(n, part), L.spx_vec_create(n, part)
    xa = np.ctypeslib.as_array(xv.contents.elements, shape=(n,))
    ya = np.ctypeslib.as_array(yv.contents.elements, shape=(n,))
    for rep in range(2):
        ya[:] = np.nan
        assert L.spx_matvec_mult(C.c_double(0.5), C.c_void_p(A.handle), xv, yv) == 0
        check_y(csr, xa.copy(), ya.copy(), 0.5)
        # (library vectors: x is resident after the first call unless spx.vec.device is off and it is sent again;
        # either way, when it was sent to a cut general stream it went by need)
        o = host_order(A)
        assert o == [] or (sorted(o) == list(range(len(o))) and o == orders.get(name, o)), (name, rep, o, orders.get(name))
    # a changed x is seen (the pieces are sent again, or the resident copy is refreshed)
    xa[:] = synth.random_x(n, seed=11)
    L.spx_hip_vec_touch(xv)
    ya[:] = np.nan
    assert L.spx_matvec_mult(C.c_double(0.5), C.c_void_p(A.handle), xv, yv) == 0
    check_y(csr, xa.copy(), ya.copy(), 0.5)
    ya[:] = y0
    L.spx_matvec_kernel.argtypes = [C.c_double, C.c_void_p, C.POINTER(VectorStruct), C.c_double, C.POINTER(VectorStruct)]
    assert L.spx_matvec_kernel(2.0, C.c_void_p(A.handle), xv, -0.5, yv) == 0
    check_y(csr, xa.copy(), ya.copy(), 2.0, -0.5, y0)
    L.spx_vec_destroy(xv); L.spx_vec_destroy(yv)
    A.destroy()
    sx.options_reset()
    print("ok", name)
'''


@pytest.mark.parametrize("min_bytes,xpiece", [("1024", "8192"), ("1024", "0"), ("1000000000000", "8192")])
def test_host_vectors_with_and_without_parts(min_bytes, xpiece):
    env = dict(os.environ, SPX_HOST_PARTS_MIN_BYTES=min_bytes)
    if xpiece != "0":
        env["SPX_HOST_XPIECE_BYTES"] = xpiece
    code = CHILD % {"root": ROOT, "tests": os.path.join(ROOT, "tests"), "expect_parts": str(min_bytes == "1024")}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("ok ") == 8
