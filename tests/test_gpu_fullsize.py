"""The BASELINE.json configurations at their full sizes on the GPU (synthetic
stand-ins of SURVEY.md section 8d).  Besides the CSR comparison (scipy's CSR
product is cheap even at 28 M nonzeros) the size-independent properties of the
operation are checked: linearity, agreement of the general and the symmetric
path, symmetry of the bilinear form, the beta path, and a checksum through the
column sums."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import sparsex_amd as sx
from sparsex_amd import synth
from helpers import tune, check_y, FP64_BOUND_FACTOR

pytestmark = pytest.mark.gpu

FULL = [
    ("syn-cant", lambda: synth.syn_cant(1.0), True),
    ("syn-nd24k", lambda: synth.syn_nd24k(1.0), True),
    ("syn-webbase", lambda: synth.syn_webbase(1.0), False),
    # BASELINE config 4's stand-in at an eighth of its nonzeros (95 M, 760 MB of values: beyond the
    # Infinity Cache) ...
    ("syn-nlpkkt-e120", lambda: synth.syn_nlpkkt_rows(120), True),
    # rounds 1-2's matrix (runs of six columns): 77 M nonzeros
    ("syn-kkt2f-e90", lambda: synth.syn_kkt2f_rows(90), True),
    # ... and at its full size, the bench matrix itself: 27 993 600 rows, 769 M nonzeros (a minute and a
    # half: generation, CSR product, tuning of both paths; SPX_TEST_SKIP_BENCH_SIZE=1 leaves it out)
] + ([] if os.environ.get("SPX_TEST_SKIP_BENCH_SIZE") == "1" else [("syn-nlpkkt-e240", lambda: synth.syn_nlpkkt_rows(240), True)])


@pytest.fixture(scope="module", params=FULL, ids=[f[0] for f in FULL])
def case(request):
    name, gen, symmetric = request.param
    csr = gen()
    rp, ci, va, n = csr
    a = sp.csr_matrix((va, ci, rp), shape=(n, n))
    A = tune(csr, {"spx.rt.nr_threads": "32" if rp[-1] > 50_000_000 else "8", "spx.rt.keep_encoded": "false"})
    return name, csr, a, A, symmetric


def mult(A, alpha, x, beta=None, y0=None):
    y = np.full(x.size, np.nan) if y0 is None else y0.copy()
    if beta is None:
        A.matvec_mult(alpha, x, y)
    else:
        A.matvec_kernel(alpha, x, beta, y)
    return y


def bound(a, x, alpha=1.0):
    return FP64_BOUND_FACTOR * 2.0 ** -53 * abs(alpha) * (abs(a) @ np.abs(x))


def test_full_size_against_csr_and_properties(case):
    name, csr, a, A, symmetric = case
    n = csr[3]
    x1, x2 = synth.random_x(n), synth.random_x(n, seed=11)
    y1, y2 = mult(A, 0.5, x1), mult(A, 0.5, x2)
    check_y(csr, x1, y1, 0.5)
    check_y(csr, x2, y2, 0.5)
    # linearity: A(2 x1 - 3 x2) = 2 A x1 - 3 A x2 within the fp64 bound of the three products
    y12 = mult(A, 0.5, 2.0 * x1 - 3.0 * x2)
    tol = 2.0 * bound(a, x1, 0.5) * 2 + 3.0 * bound(a, x2, 0.5) * 2 + bound(a, 2.0 * x1 - 3.0 * x2, 0.5)
    assert np.all(np.abs(y12 - (2.0 * y1 - 3.0 * y2)) <= tol + 1e-300)
    # checksum of checksums: sum(y) = alpha * (1^T A) x through the column sums
    colsum = np.asarray(a.sum(axis=0)).ravel()
    s_ref = 0.5 * float(colsum @ x1)
    s_tol = float(bound(a, x1, 0.5).sum()) + 64 * 2.0 ** -53 * float(np.abs(colsum) @ np.abs(x1))
    assert abs(float(y1.sum()) - s_ref) <= s_tol
    # beta path at full size
    y0 = synth.random_x(n, seed=5)
    yk = mult(A, -1.25, x1, 0.75, y0)
    check_y(csr, x1, yk, -1.25, 0.75, y0)


def test_full_size_symmetric_path_agrees_with_general(case):
    name, csr, a, A, symmetric = case
    if not symmetric:
        pytest.skip("unsymmetric matrix")
    n = csr[3]
    S = tune(csr, {"spx.rt.nr_threads": "8", "spx.rt.keep_encoded": "false"}, sym=True)
    # beyond 16 M nonzeros in the triangle, runs of consecutive columns are read once
    assert (S.info().sym_segments > 0) == name.startswith(("syn-nlpkkt-e", "syn-kkt2f-e"))
    x, z = synth.random_x(n), synth.random_x(n, seed=23)
    yg, ys = mult(A, 0.5, x), mult(S, 0.5, x)
    check_y(csr, x, ys, 0.5)
    assert np.all(np.abs(yg - ys) <= 2.0 * bound(a, x, 0.5) + 1e-300)
    # symmetry of the bilinear form: z^T (A x) = x^T (A z)
    zs = mult(S, 0.5, z)
    lhs, rhs = float(z @ ys), float(x @ zs)
    tol = float(np.abs(z) @ bound(a, x, 0.5) + np.abs(x) @ bound(a, z, 0.5)) + 1e-300
    assert abs(lhs - rhs) <= tol


def test_full_size_library_vectors(case):
    """spx_matvec_mult / spx_matvec_kernel on vectors the library created (page-locked): on the bench matrix x goes
    up piece by piece in the order the parts of the product need it while finished rows of y come back, and with
    beta != 0 a part's rows of y go up the same way (device_spmv_host) -- general and symmetric path against CSR.
    A product with another x comes first: a piece the plan forgot would still hold it."""
    import ctypes as C
    from sparsex_amd.api import VectorStruct
    name, csr, a, A, symmetric = case
    n = csr[3]
    L = sx.lib()
    L.spx_vec_create_random.restype = C.POINTER(VectorStruct); L.spx_vec_create_random.argtypes = [C.c_size_t, C.c_void_p]
    L.spx_vec_create.restype = C.POINTER(VectorStruct); L.spx_vec_create.argtypes = [C.c_size_t, C.c_void_p]
    L.spx_mat_get_partition.restype = C.c_void_p
    L.spx_matvec_kernel.argtypes = [C.c_double, C.c_void_p, C.POINTER(VectorStruct), C.c_double, C.POINTER(VectorStruct)]
    L.spx_hip_mat_host_parts.restype = C.c_int
    L.spx_hip_mat_host_order.restype = C.c_int
    L.spx_hip_mat_host_order.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int]
    mats = [A] + ([tune(csr, {"spx.rt.nr_threads": "8", "spx.rt.keep_encoded": "false"}, sym=True)] if symmetric else [])
    for M in mats:
        part = C.c_void_p(L.spx_mat_get_partition(C.c_void_p(M.handle)))
        xv, yv = L.spx_vec_create_random(n, part), L.spx_vec_create(n, part)
        xa = np.ctypeslib.as_array(xv.contents.elements, shape=(n,))
        ya = np.ctypeslib.as_array(yv.contents.elements, shape=(n,))
        x1, y0 = synth.random_x(n, seed=31), synth.random_x(n, seed=37)
        xa[:] = np.nan
        assert L.spx_matvec_mult(C.c_double(0.5), C.c_void_p(M.handle), xv, yv) == 0
        xa[:] = x1
        ya[:] = np.nan
        assert L.spx_matvec_mult(C.c_double(0.5), C.c_void_p(M.handle), xv, yv) == 0
        check_y(csr, x1, ya.copy(), 0.5)
        parts = L.spx_hip_mat_host_parts(C.c_void_p(M.handle))
        buf = (C.c_int32 * 64)()
        k = L.spx_hip_mat_host_order(C.c_void_p(M.handle), buf, 64)
        if n * 8 >= (32 << 20) and (M is A or name.startswith("syn-nlpkkt-e")):
            # (the bench matrix: cut on both paths -- its symmetric stream's row-blocks all store their own rows)
            assert parts >= 2 and sorted(buf[i] for i in range(k)) == list(range(parts)), (name, parts, k)
        ya[:] = y0
        assert L.spx_matvec_kernel(-1.25, C.c_void_p(M.handle), xv, 0.75, yv) == 0
        check_y(csr, x1, ya.copy(), -1.25, 0.75, y0)
        L.spx_vec_destroy(xv); L.spx_vec_destroy(yv)
        # views of client arrays kept across products (the reference harness' pattern): from 32 MB on they are
        # page-locked where they lie at the first product and travel like the library's own vectors
        L.spx_vec_create_from_buff.restype = C.POINTER(VectorStruct)
        L.spx_vec_create_from_buff.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.spx_hip_vec_page_locked.restype = C.c_int
        L.spx_hip_vec_page_locked.argtypes = [C.POINTER(VectorStruct)]
        xb, yb = x1.copy(), np.full(n, np.nan)
        xw = L.spx_vec_create_from_buff(xb.ctypes.data, None, n, None, 43)
        yw = L.spx_vec_create_from_buff(yb.ctypes.data, None, n, None, 43)
        assert L.spx_matvec_mult(C.c_double(0.5), C.c_void_p(M.handle), xw, yw) == 0
        check_y(csr, x1, yb.copy(), 0.5)
        locked = L.spx_hip_vec_page_locked(xw)
        assert locked in ((0, 2, 3) if n * 8 >= (32 << 20) else (0,)), (name, locked)
        if locked and L.spx_hip_mat_host_parts(C.c_void_p(M.handle)) >= 2:
            k = L.spx_hip_mat_host_order(C.c_void_p(M.handle), buf, 64)
            assert sorted(buf[i] for i in range(k)) == list(range(L.spx_hip_mat_host_parts(C.c_void_p(M.handle)))), (name, k)
        yb[:] = y0
        assert L.spx_matvec_kernel(-1.25, C.c_void_p(M.handle), xw, 0.75, yw) == 0
        check_y(csr, x1, yb.copy(), -1.25, 0.75, y0)
        L.spx_vec_destroy(xw); L.spx_vec_destroy(yw)
        if M is not A:
            M.destroy()
