"""Randomised check of the whole host pipeline (preprocessor -> row-block stream):
matrices with random mixtures of structure, random options; the saved stream,
decoded lane by lane by tests/stream_decode.py, must hold exactly the input
matrix, and its emulated product must equal the CSR product."""
import numpy as np
import pytest
import scipy.sparse as sp

import sparsex_amd as sx
from sparsex_amd import synth
from helpers import tune
from stream_decode import Stream


def random_matrix(seed, symmetric):
    rng = np.random.RandomState(seed)
    n = int(rng.choice([40, 97, 256, 700, 1500]))
    rows, cols = [np.arange(n)], [np.arange(n)]
    for _ in range(rng.randint(1, 7)):
        kind = rng.randint(0, 6)
        if kind == 0:                                   # random scatter
            k = rng.randint(1, 4 * n)
            rows.append(rng.randint(0, n, k)); cols.append(rng.randint(0, n, k))
        elif kind == 1:                                 # diagonals with strides
            off, st = rng.randint(-n // 2, n // 2), rng.randint(1, 4)
            r = np.arange(max(0, -off), min(n, n - off), st)
            rows.append(r); cols.append(r + off)
        elif kind == 2:                                 # dense blocks
            for _ in range(rng.randint(1, 12)):
                h, w = rng.randint(1, 17), rng.randint(1, 25)
                r0, c0 = rng.randint(0, n - h + 1) if n > h else 0, rng.randint(0, max(1, n - w + 1))
                a, b = np.meshgrid(np.arange(min(h, n)), np.arange(min(w, n)), indexing="ij")
                rows.append(np.minimum(r0 + a.ravel(), n - 1)); cols.append(np.minimum(c0 + b.ravel(), n - 1))
        elif kind == 3:                                 # stencil: a few fixed offsets for every row
            for off in rng.randint(-30, 31, rng.randint(2, 7)):
                r = np.arange(max(0, -off), min(n, n - off))
                keep = rng.rand(r.size) > 0.05
                rows.append(r[keep]); cols.append(r[keep] + off)
        elif kind == 4:                                 # a few very long rows and columns
            for _ in range(rng.randint(1, 3)):
                r = rng.randint(0, n)
                c = rng.choice(n, rng.randint(n // 3, n), replace=False)
                rows.append(np.full(c.size, r)); cols.append(c)
        else:                                           # aligned 8x8 tiles (what the symmetric path reads once)
            nb = n // 8
            for _ in range(rng.randint(1, 3 * nb + 2)):
                i, j = rng.randint(0, nb), rng.randint(0, nb)
                a, b = np.meshgrid(np.arange(8), np.arange(8), indexing="ij")
                rows.append(8 * i + a.ravel()); cols.append(8 * j + b.ravel())
    r, c = np.concatenate(rows), np.concatenate(cols)
    if symmetric:
        r, c = np.concatenate([r, c]), np.concatenate([c, r])
    m = sp.coo_matrix((np.ones(r.size), (r, c)), shape=(n, n)).tocsr()
    m.sum_duplicates(); m.sort_indices()
    m.data = rng.uniform(0.5, 1.5, m.nnz)
    if symmetric:
        low = sp.tril(m, k=-1)
        m = (low + low.T + sp.diags(m.diagonal())).tocsr()
        m.sort_indices()
    return (m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.copy(), n), m


def random_options(seed):
    rng = np.random.RandomState(1000 + seed)
    o = {"spx.rt.nr_threads": str(rng.choice([1, 2, 3, 5]))}
    if rng.rand() < 0.7:
        o["spx.preproc.sampling"] = "none"
    if rng.rand() < 0.5:
        o["spx.preproc.xform"] = str(rng.choice(["all", "h", "v", "d", "ad", "br", "bc", "h,d", "v,br"]))
    if rng.rand() < 0.5:
        o["spx.gpu.rowblock_elems"] = str(rng.choice([64, 200, 1000, 4096]))
        o["spx.gpu.rowblock_rows"] = str(rng.choice([3, 16, 100, 512]))
    for k in ("spx.gpu.stack_segments", "spx.gpu.recut_linear"):
        if rng.rand() < 0.25:
            o[k] = "false"
    return o


def random_sym_options(seed, o):
    """... plus the choices of the symmetric path: mirrored / tiles / read-once segments."""
    rng = np.random.RandomState(seed)
    for k in ("spx.gpu.sym_once", "spx.gpu.sym_remine"):
        if rng.rand() < 0.3:
            o[k] = "false"
    if rng.rand() < 0.6:
        o["spx.gpu.sym_segments"] = "true"
        o["spx.gpu.sym_wide_rows"] = str(rng.choice([512, 700, 1024, 2048]))
        o["spx.gpu.sym_segment_min"] = str(rng.choice([2, 3, 4]))
    # round 6 (a generator of its own, so that the draws above stay what they were): the read-once pipeline and
    # the passes of their own that feed it
    r6 = np.random.RandomState(600000 + seed)
    o["spx.gpu.sym_pipeline"] = str(r6.choice(["true", "auto", "false"]))
    if r6.rand() < 0.3:
        o["spx.gpu.sym_pure_passes"] = "false"
    return o


@pytest.mark.parametrize("seed", range(40))
def test_general_stream_random(tmp_path, seed):
    csr, m = random_matrix(seed, symmetric=False)
    rp, ci, va, n = csr
    A = tune(csr, random_options(seed), host_only=True)
    f = str(tmp_path / "m.spx")
    A.save(f)
    s = Stream(f)
    r, c, v, b = s.triplets()
    assert r.size == m.nnz and np.unique(r * n + c).size == r.size
    got = sp.coo_matrix((v, (r, c)), shape=(n, n)).tocsr()
    assert abs(got - m).max() == 0
    s.check_ownership()
    x = synth.random_x(n)
    assert np.allclose(s.matvec(x), m @ x, rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("seed", range(40, 90))
def test_symmetric_stream_random(tmp_path, seed):
    csr, m = random_matrix(seed, symmetric=True)
    rp, ci, va, n = csr
    o = random_sym_options(seed, random_options(seed))
    A = tune(csr, o, sym=True, host_only=True)
    f = str(tmp_path / "m.spx")
    A.save(f)
    s = Stream(f)
    r, c, v, b = s.triplets()
    got = sp.coo_matrix((v, (r, c)), shape=(n, n)).tocsr()      # explicit zeros on the diagonal sum away
    off = (m - sp.diags(m.diagonal())).tocsr()
    assert abs(got - off).max() == 0
    assert np.array_equal(s.dvalues, m.diagonal())
    s.check_ownership()
    x = synth.random_x(n)
    assert np.allclose(s.matvec(x), m @ x, rtol=1e-12, atol=1e-13)
