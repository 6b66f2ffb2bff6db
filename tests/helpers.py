"""Shared helpers of the test-suite."""
import os

import numpy as np

import sparsex_amd as sx
from sparsex_amd import synth
from oracle import pyoracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# |y_i - yhat_i| <= FP64_BOUND_FACTOR * 2^-53 * sum_j |a_ij||x_j|: summation-order
# independent bound for a dot product of up to ~2^5 * (row length) roundings;
# the stated fp64 tolerance of this build
FP64_BOUND_FACTOR = 64.0


def tune(csr, opts=None, sym=False, host_only=False):
    rp, ci, va, n = csr
    sx.options_reset()
    if host_only:
        sx.option_set("spx.rt.host_only", "true")
    for k, v in (opts or {}).items():
        sx.option_set(k, v)
    if sym:
        sx.option_set("spx.matrix.symmetric", "true")
    inp = sx.input_load_csr(rp, ci, va, n, n)
    A = sx.mat_tune(inp)
    A._input = inp
    return A


def oracle_y(A, x, alpha=1.0, nthreads=1):
    inf = A.info()
    ex = [A.export_csx(p) for p in range(inf.first_partition, inf.last_partition)]
    P = pyoracle.Partitions(ex, bool(inf.symmetric))
    return pyoracle.csx_matvec(P, x, A.nrows, alpha, nthreads), ex


def abs_bound(csr, x, alpha=1.0):
    rp, ci, va, n = csr
    import scipy.sparse as sp
    absA = sp.csr_matrix((np.abs(va), ci, rp), shape=(n, n))
    return abs(alpha) * FP64_BOUND_FACTOR * 2.0 ** -53 * (absA @ np.abs(x)) + 1e-300


def check_y(csr, x, y, alpha=1.0, beta=0.0, y0=None):
    """Reference criterion (relative 1e-6 vs CSR) and the fp64 bound."""
    rp, ci, va, n = csr
    yc = alpha * pyoracle.csr_matvec(rp, ci, va, x)
    bound = abs_bound(csr, x, alpha)
    if beta != 0.0:
        yc = yc + beta * y0
        bound = bound + 4 * 2.0 ** -53 * np.abs(beta * y0)
    assert pyoracle.vec_compare(yc, y) == 0, "relative 1e-6 criterion failed"
    err = np.abs(y - yc)
    assert np.all(err <= bound), "fp64 bound exceeded: max ratio %g" % (err / bound).max()


def check_vs_oracle(csr, x, y, yo, alpha=1.0):
    """The HIP product against the oracle's product of the SAME tuned matrix, directly:
    both lie within the fp64 bound of the exact product, so they differ by at most
    twice that bound; and the reference's own relative 1e-6 criterion."""
    assert pyoracle.vec_compare(yo, y) == 0, "HIP vs oracle: relative 1e-6 criterion failed"
    err = np.abs(y - yo)
    bound = 2.0 * abs_bound(csr, x, alpha)
    assert np.all(err <= bound), "HIP vs oracle: max ratio %g" % (err / bound).max()
