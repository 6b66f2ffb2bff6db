"""spx_matvec_kernel_csr (reference src/api/matvec.c:622-673): y <- alpha*A*x + beta*y
straight from CSR arrays -- the first call tunes and hands back the handle, later
calls only multiply."""
import numpy as np
import pytest

import sparsex_amd as sx
from sparsex_amd import synth
from helpers import check_y


def test_argument_checks():
    """The reference's checks, in its order: x, y, the handle pointer; for a fresh handle the
    dimensions and the three arrays (matvec.c:628-652).  No GPU needed to be refused."""
    rp, ci, va, n = synth.syn_cant(0.01)
    x, y = synth.random_x(n), np.zeros(n)
    sx.lib().spx_log_disable_all()
    with pytest.raises(sx.SpxError):
        sx.matvec_kernel_csr(None, rp, ci, va, n, n, 1.0, None, 0.0, y)
    with pytest.raises(sx.SpxError):
        sx.matvec_kernel_csr(None, rp, ci, va, n, n, 1.0, x, 0.0, None)
    with pytest.raises(sx.SpxError):
        sx.matvec_kernel_csr(None, rp, ci, va, 0, n, 1.0, x, 0.0, y)
    with pytest.raises(sx.SpxError):
        sx.matvec_kernel_csr(None, rp, ci, va, n, -3, 1.0, x, 0.0, y)
    for missing in range(3):
        arrs = [rp, ci, va]
        arrs[missing] = None
        with pytest.raises(sx.SpxError):
            sx.matvec_kernel_csr(None, arrs[0], arrs[1], arrs[2], n, n, 1.0, x, 0.0, y)


def test_host_only_handle_cannot_multiply():
    rp, ci, va, n = synth.syn_cant(0.01)
    sx.option_set("spx.rt.host_only", "true")
    sx.lib().spx_log_disable_all()
    with pytest.raises(sx.SpxError):                 # tuned, but there is no CPU multiplication path
        sx.matvec_kernel_csr(None, rp, ci, va, n, n, 1.0, synth.random_x(n), 0.0, np.zeros(n))


@pytest.mark.gpu
@pytest.mark.parametrize("gen", [lambda: synth.syn_cant(0.05), lambda: synth.syn_webbase(0.02),
                                 lambda: synth.syn_nlpkkt(8)], ids=["cant", "webbase", "nlpkkt"])
def test_first_call_tunes_later_calls_multiply(gen):
    csr = gen()
    rp, ci, va, n = csr
    x = synth.random_x(n)
    y0 = synth.random_x(n, seed=7)
    y = y0.copy()
    A = sx.matvec_kernel_csr(None, rp, ci, va, n, n, 0.8, x, 0.42, y)       # alpha/beta of advanced_example.c
    check_y(csr, x, y, 0.8, 0.42, y0)
    assert (A.nrows, A.ncols, A.nnz) == (n, n, int(rp[-1]))
    # the handle is reused: the arrays are not looked at again
    y1 = y.copy()
    B = sx.matvec_kernel_csr(A, None, None, None, 0, 0, 0.8, x, 0.42, y)
    assert B is A
    check_y(csr, x, y, 0.8, 0.42, y1)
    y2 = np.full(n, np.nan)
    sx.matvec_kernel_csr(A, None, None, None, 0, 0, 2.0, x, 0.0, y2)          # beta = 0 ignores y
    check_y(csr, x, y2, 2.0)
    # and it is an ordinary matrix handle
    y3 = np.zeros(n)
    A.matvec_mult(2.0, x, y3)
    assert np.array_equal(y2, y3) or np.allclose(y2, y3, rtol=1e-13, atol=1e-15)
