"""Edge cases the domain has: rectangular matrices, empty matrix, leading /
trailing / interior empty rows, a single dense row, one-column matrix."""
import numpy as np
import pytest
import scipy.sparse as sp

import sparsex_amd as sx
from helpers import oracle_y
from oracle import pyoracle


def _csr(a):
    a = sp.csr_matrix(a)
    a.sort_indices()
    return a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data.astype(np.float64)


def _tune(a, opts, host_only):
    rp, ci, va = _csr(a)
    sx.options_reset()
    if host_only:
        sx.option_set("spx.rt.host_only", "true")
    for k, v in opts.items():
        sx.option_set(k, v)
    inp = sx.input_load_csr(rp, ci, va, a.shape[0], a.shape[1])
    A = sx.mat_tune(inp)
    A._input = inp
    return A


def _matrices():
    rng = np.random.RandomState(11)
    out = {}
    out["wide"] = sp.random(60, 400, density=0.05, random_state=rng, format="csr")
    out["tall"] = sp.random(500, 40, density=0.08, random_state=rng, format="csr")
    m = sp.random(300, 300, density=0.03, random_state=rng, format="lil")
    m[:40, :] = 0
    m[120:180, :] = 0
    m[260:, :] = 0
    out["empty_rows"] = m.tocsr()
    d = sp.lil_matrix((50, 2000))
    d[7, :] = rng.uniform(-1, 1, 2000)
    d[20, 5] = 1.0
    out["one_dense_row"] = d.tocsr()
    out["one_column"] = sp.csr_matrix(rng.uniform(-1, 1, (200, 1)))
    band = sp.diags([np.ones(399), 2 * np.ones(400), np.ones(399)], [-1, 0, 1], format="csr")
    out["tridiagonal_rect"] = sp.hstack([band, sp.csr_matrix((400, 37))]).tocsr()
    return out


MATS = _matrices()
OPTS = [{}, {"spx.preproc.sampling": "none"}, {"spx.preproc.sampling": "none",
                                               "spx.rt.nr_threads": "3"}]


@pytest.mark.parametrize("name", sorted(MATS))
@pytest.mark.parametrize("opts", OPTS)
def test_host_streams_decode_to_csr_product(name, opts):
    a = MATS[name]
    a.eliminate_zeros()
    A = _tune(a, opts, host_only=True)
    x = np.random.RandomState(1).uniform(-0.1, 0.1, a.shape[1])
    inf = A.info()
    ex = [A.export_csx(p) for p in range(inf.nr_partitions)]
    y = pyoracle.csx_matvec(pyoracle.Partitions(ex, False), x, a.shape[0], 0.5)
    yc = 0.5 * (a @ x)
    assert np.allclose(y, yc, rtol=1e-12, atol=1e-14)
    assert inf.nnz_stored == a.nnz


def test_empty_matrix_host():
    a = sp.csr_matrix((30, 20))
    A = _tune(a, {}, host_only=True)
    assert A.nnz == 0 and A.info().nnz_stored == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(MATS))
@pytest.mark.parametrize("opts", OPTS)
def test_gpu_rectangular_and_empty_rows(name, opts):
    a = MATS[name]
    a.eliminate_zeros()
    A = _tune(a, opts, host_only=False)
    x = np.random.RandomState(1).uniform(-0.1, 0.1, a.shape[1])
    y = np.full(a.shape[0], np.nan)
    A.matvec_mult(0.5, x, y)
    yc = 0.5 * (a @ x)
    assert np.allclose(y, yc, rtol=1e-12, atol=1e-14)
    y0 = np.random.RandomState(2).uniform(-1, 1, a.shape[0])
    y = y0.copy()
    A.matvec_kernel(2.0, x, -0.5, y)
    assert np.allclose(y, 2.0 * (a @ x) - 0.5 * y0, rtol=1e-12, atol=1e-14)


@pytest.mark.gpu
def test_gpu_empty_matrix():
    a = sp.csr_matrix((30, 20))
    A = _tune(a, {}, host_only=False)
    y = np.full(30, np.nan)
    A.matvec_mult(1.0, np.ones(20), y)
    assert not y.any()
    y = np.arange(30, dtype=np.float64)
    A.matvec_kernel(1.0, np.ones(20), 3.0, y)
    assert np.array_equal(y, 3.0 * np.arange(30))


def test_malformed_row_pointers_do_not_read_behind_colind():
    """Row pointers that step back ([0, 1000000, 10]: ten elements by the last pointer, which is how the C API
    learns the element count, Csr.hpp:66) must be refused before anything is read at colind[1000000]."""
    import ctypes as C
    L = sx.lib()
    rp = np.array([0, 1000000, 10], dtype=np.int32)
    ci = np.arange(10, dtype=np.int32)
    va = np.ones(10)
    sx.options_reset()
    sx.option_set("spx.rt.host_only", "true")
    inp = L.spx_input_load_csr(rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p),
                               va.ctypes.data_as(C.c_void_p), C.c_int(2), C.c_int(2000000), C.c_int(0))
    assert not inp
    # the partition-aware renumbering validates its arrays on every path
    for sym in (False, True):
        with pytest.raises(sx.SpxError):
            sx.dist_reorder(rp, ci, 2, 2, pattern_symmetric=sym)
    sx.options_reset()
