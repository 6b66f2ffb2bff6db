"""spx_mat_get_entry / spx_mat_set_entry (reference: src/api/matvec.c:324-407,
CsxGetSet.hpp): random access into the tuned matrix, whatever unit a nonzero
ended up in; a changed value shows in the next product."""
import numpy as np
import pytest
import scipy.sparse as sp

import sparsex_amd as sx
from sparsex_amd import synth
from helpers import tune, oracle_y, check_y


@pytest.mark.parametrize("gen,opts,sym", [
    (lambda: synth.syn_cant(0.02), {"spx.preproc.sampling": "none"}, False),
    (lambda: synth.syn_nlpkkt(6), {"spx.preproc.sampling": "none", "spx.rt.nr_threads": "3"}, False),
    (lambda: synth.syn_nd24k(0.012), {"spx.preproc.sampling": "none"}, False),
    (lambda: synth.syn_webbase(0.004), {}, False),
    (lambda: synth.syn_cant(0.02), {"spx.preproc.sampling": "none", "spx.rt.nr_threads": "2"}, True),
])
def test_get_every_entry_and_set_some_host(gen, opts, sym):
    csr = gen()
    rp, ci, va, n = csr
    A = tune(csr, opts, sym=sym, host_only=True)
    rng = np.random.RandomState(0)
    rows = np.repeat(np.arange(n), np.diff(rp))
    pick = rng.choice(rp[-1], size=min(3000, rp[-1]), replace=False)
    for j in pick:
        assert A.get_entry(int(rows[j]), int(ci[j])) == va[j]
        assert A.get_entry(int(rows[j]) + 1, int(ci[j]) + 1, sx.SPX_INDEX_ONE_BASED) == va[j]
    sx.lib().spx_log_disable_all()
    a = sp.csr_matrix((va, ci, rp), shape=(n, n))
    zr, zc = 0, n - 1
    if a[zr, zc] == 0:
        with pytest.raises(sx.SpxError):
            A.get_entry(zr, zc)
        with pytest.raises(sx.SpxError):
            A.set_entry(zr, zc, 1.0)
    with pytest.raises(sx.SpxError):
        A.get_entry(n, 0)                         # out of bounds
    # overwrite a few values; the exported stream and its product follow
    va2 = va.copy()
    for j in pick[:40]:
        r, c = int(rows[j]), int(ci[j])
        newv = float(rng.uniform(-2, 2))
        A.set_entry(r, c, newv)
        va2[j] = newv
        if sym:                                    # the mirrored entry is the same storage
            k = rp[c] + int(np.searchsorted(ci[rp[c]:rp[c + 1]], r))
            va2[k] = newv
    x = synth.random_x(n)
    yo, _ = oracle_y(A, x, 1.0)
    check_y((rp, ci, va2, n), x, yo, 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("sym", [False, True])
def test_set_entry_reaches_the_gpu(sym):
    csr = synth.syn_cant(0.03)
    rp, ci, va, n = csr
    A = tune(csr, {"spx.preproc.sampling": "none"}, sym=sym)
    x = synth.random_x(n)
    y = np.zeros(n)
    A.matvec_mult(1.0, x, y)
    check_y(csr, x, y, 1.0)
    rows = np.repeat(np.arange(n), np.diff(rp))
    rng = np.random.RandomState(5)
    va2 = va.copy()
    for j in rng.choice(rp[-1], size=25, replace=False):
        r, c = int(rows[j]), int(ci[j])
        A.set_entry(r, c, 3.5)
        va2[j] = 3.5
        if sym:
            va2[rp[c] + int(np.searchsorted(ci[rp[c]:rp[c + 1]], r))] = 3.5
    y2 = np.zeros(n)
    A.matvec_mult(1.0, x, y2)
    check_y((rp, ci, va2, n), x, y2, 1.0)
    assert not np.allclose(y, y2)
