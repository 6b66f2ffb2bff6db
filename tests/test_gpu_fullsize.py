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
