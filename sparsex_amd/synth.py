"""Deterministic synthetic stand-ins for the SuiteSparse matrices named in
BASELINE.json (the real files are not available offline).  Structure follows
SURVEY.md section 8(d): same dimensions, nonzeros per row and pattern classes
as cant / nd24k / webbase-1M / nlpkkt, so that the same CSX unit types
dominate.  All generators return zero-based CSR arrays
``(rowptr int32, colind int32 (sorted per row), values float64, n)``.

Host-side input generation only -- nothing here takes part in the SpMV.
"""
import numpy as np
import scipy.sparse as sp

SEED_BASE = 0x5EED0000


def _finish(coo_rows, coo_cols, n, rng, symmetric):
    """Unique pattern -> CSR with values U(-1,1) (symmetric if requested)."""
    a = sp.coo_matrix((np.ones(coo_rows.size, dtype=np.int8), (coo_rows, coo_cols)),
                      shape=(n, n)).tocsr()
    a.sum_duplicates()
    a.sort_indices()
    if symmetric:
        low = sp.tril(a, k=-1, format="coo")
        vals = rng.uniform(-1.0, 1.0, low.nnz)
        lowv = sp.coo_matrix((vals, (low.row, low.col)), shape=(n, n))
        full = (lowv + lowv.T).tocsr()
        # full, strictly positive diagonal (the symmetric path requires one)
        rowsum = np.asarray(abs(full).sum(axis=1)).ravel()
        full = (full + sp.diags(rowsum + 1.0)).tocsr()
        full.sort_indices()
        a = full
    else:
        a = a.astype(np.float64)
        a.data = rng.uniform(-1.0, 1.0, a.nnz)
    return (a.indptr.astype(np.int32), a.indices.astype(np.int32),
            a.data.astype(np.float64), n)


def syn_cant(scale=1.0, seed=SEED_BASE + 1):
    """FEM-like banded matrix: 3-dof nodes, ~21 coupled nodes within +-600,
    ~63 nnz/row (cant: 62451 rows, ~4.0M nnz, symmetric)."""
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    nodes = max(64, int(20817 * scale))
    n = nodes * 3
    band = max(8, int(600 * min(1.0, scale * 4)))
    # regular mesh offsets (a few short ones and mesh-line strides), symmetric
    base = np.unique(np.concatenate([[1, 2, 3], rng.randint(4, band, 7)]))
    i = np.arange(nodes)
    rows, cols = [i], [i]
    for s in base:
        keep = rng.rand(nodes) > 0.03          # irregularity: dropped couplings
        src = i[keep & (i + s < nodes)]
        rows += [src, src + s]
        cols += [src + s, src]
    nr = np.concatenate(rows)
    nc = np.concatenate(cols)
    # expand nodes to 3x3 dof blocks
    a, b = np.meshgrid(np.arange(3), np.arange(3), indexing="ij")
    r = (nr[:, None] * 3 + a.ravel()[None, :]).ravel()
    c = (nc[:, None] * 3 + b.ravel()[None, :]).ravel()
    return _finish(r, c, n, rng, symmetric=True)


def syn_nd24k(scale=1.0, seed=SEED_BASE + 2):
    """3-D mesh-like matrix of dense 8x8 blocks, ~50 blocks per block row
    inside a +-6000 band, ~400 nnz/row (nd24k: 72000 rows, 28.7M nnz)."""
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    nb = max(32, int(9000 * scale))
    n = nb * 8
    band = max(30, int(750 * min(1.0, scale * 4)))
    per_row = 25 if scale >= 0.05 else 8
    i = np.repeat(np.arange(nb), per_row)
    off = rng.randint(1, band, i.size)
    j = i + off
    ok = j < nb
    i, j = i[ok], j[ok]
    br = np.concatenate([np.arange(nb), i, j])
    bc = np.concatenate([np.arange(nb), j, i])
    a, b = np.meshgrid(np.arange(8), np.arange(8), indexing="ij")
    r = (br[:, None] * 8 + a.ravel()[None, :]).ravel()
    c = (bc[:, None] * 8 + b.ravel()[None, :]).ravel()
    return _finish(r, c, n, rng, symmetric=True)


def syn_webbase(scale=1.0, seed=SEED_BASE + 3):
    """Power-law web graph: Zipf row lengths (mean ~3.1, max 4700), 70 % of
    the links uniform, 30 % near the diagonal, ~15 % empty rows, unsymmetric
    (webbase-1M: 1000005 rows, 3.1M nnz)."""
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    n = max(256, int(1000005 * scale))
    lens = rng.zipf(2.1, n).astype(np.int64)
    lens = np.minimum(lens, min(4700, n // 2))
    lens[rng.rand(n) < 0.15] = 0
    lens = np.minimum(lens, np.where(lens > 3, lens, 3))
    rows = np.repeat(np.arange(n), lens)
    uni = rng.rand(rows.size) < 0.7
    cols = np.where(uni, rng.randint(0, n, rows.size),
                    np.clip(rows + rng.randint(-1000, 1001, rows.size), 0, n - 1))
    return _finish(rows, cols, n, rng, symmetric=False)


def syn_nlpkkt(N=12, seed=SEED_BASE + 4):
    """KKT system [H A^T; A D] from 27-point stencils on an N^3 grid
    (nlpkkt240 is N=240: 27 993 600 rows, ~760M nnz; tests use small N)."""
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    n1 = 2 * N ** 3
    n2 = 6 * N ** 2 if N >= 3 else 0
    n = n1 + n2
    idx = np.arange(N ** 3)
    z, y, x = idx // (N * N), (idx // N) % N, idx % N
    rows, cols = [], []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                zz, yy, xx = z + dz, y + dy, x + dx
                ok = (zz >= 0) & (zz < N) & (yy >= 0) & (yy < N) & (xx >= 0) & (xx < N)
                src = idx[ok]
                dst = (zz * N * N + yy * N + xx)[ok]
                # H block (two interleaved fields) and their coupling
                rows += [2 * src, 2 * src + 1, 2 * src]
                cols += [2 * dst, 2 * dst + 1, 2 * dst + 1]
    # constraint rows couple boundary faces to the fields
    if n2:
        face = np.arange(n2)
        tgt = (face * 7919) % n1
        rows += [n1 + face, tgt, n1 + face]
        cols += [tgt, n1 + face, (tgt + 1) % n1]
        rows += [(tgt + 1) % n1]
        cols += [n1 + face]
    r = np.concatenate(rows)
    c = np.concatenate(cols)
    r, c = np.concatenate([r, c]), np.concatenate([c, r])     # symmetrise pattern
    return _finish(r, c, n, rng, symmetric=True)


def syn_bandrandom(n=200000, band=300, per_row=40, seed=SEED_BASE + 9):
    """Random nonzeros inside a narrow band: no substructure for CSX to find (everything
    stays a leftover), but every row-block's columns fit a small window of x -- the case
    the LDS-staged x window is for.  Not a BASELINE configuration."""
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    rows = np.repeat(np.arange(n), per_row)
    cols = np.clip(rows + rng.randint(-band, band + 1, rows.size), 0, n - 1)
    return _finish(rows, cols, n, rng, symmetric=False)


def lower_plus_diag_nnz(rowptr, colind):
    """nnz_lower + n for a symmetric matrix given in full."""
    n = rowptr.size - 1
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    return int((colind < rows).sum()) + n


def random_x(ncols, seed=42):
    """x ~ U(-0.1, 0.1), the range of the reference's VecCreateRandom
    (src/internals/Vector.cpp:161-167)."""
    return np.random.RandomState(seed).uniform(-0.1, 0.1, ncols)


_SYNLIB = None


def _synlib():
    """libspxsynth.so (tools/synth/nlpkkt_gen.c, built by `make lib`): the
    row-sliced generator of the nlpkkt stand-in."""
    global _SYNLIB
    if _SYNLIB is None:
        import ctypes as C
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libspxsynth.so")
        L = C.CDLL(path)
        L.spx_syn_nlpkkt_nrows.restype = C.c_int64
        L.spx_syn_nlpkkt_nrows.argtypes = [C.c_int]
        L.spx_syn_nlpkkt_counts.restype = None
        L.spx_syn_nlpkkt_counts.argtypes = [C.c_int, C.c_void_p]
        L.spx_syn_nlpkkt_rows.restype = C.c_int64
        L.spx_syn_nlpkkt_rows.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_uint64, C.c_void_p,
                                          C.c_void_p, C.c_void_p]
        _SYNLIB = L
    return _SYNLIB


def nlpkkt_nrows(N):
    return int(_synlib().spx_syn_nlpkkt_nrows(int(N)))


def nlpkkt_row_counts(N):
    """Nonzeros of every row of syn_nlpkkt_rows(N) (int32, all rows)."""
    cnt = np.empty(nlpkkt_nrows(N), dtype=np.int32)
    _synlib().spx_syn_nlpkkt_counts(int(N), cnt.ctypes.data)
    return cnt


def syn_nlpkkt_rows(N, lo=0, hi=None, counts=None, seed=SEED_BASE + 4):
    """Rows [lo, hi) of the nlpkkt stand-in as a CSR slice: the pattern of
    syn_nlpkkt(N), generated row by row in C so that a process can hold only
    the rows it owns (values: symmetric hash of the coordinate pair, diagonal
    dominant).  Returns (rowptr int32 relative to the slice, colind int32,
    values float64, n) with n the order of the WHOLE matrix."""
    n = nlpkkt_nrows(N)
    hi = n if hi is None else hi
    if counts is None:
        counts = nlpkkt_row_counts(N)
    nnz = int(counts[lo:hi].sum(dtype=np.int64))
    assert nnz < 2 ** 31, "slice too large for 32-bit row pointers"
    rp = np.zeros(hi - lo + 1, dtype=np.int64)
    ci = np.zeros(nnz, dtype=np.int32)
    va = np.zeros(nnz, dtype=np.float64)
    got = _synlib().spx_syn_nlpkkt_rows(int(N), int(lo), int(hi), int(seed), rp.ctypes.data,
                                        ci.ctypes.data, va.ctypes.data)
    assert got == nnz
    return rp.astype(np.int32), ci, va, n


def nlpkkt_edge(scale):
    """Grid edge for a size factor: scale 1 is nlpkkt240's order (N = 240)."""
    return max(3, int(round(240.0 * scale ** (1.0 / 3.0))))


def syn_nlpkkt_scaled(scale=1.0):
    """syn_nlpkkt with the grid edge derived from a size factor: scale 1 is
    nlpkkt240 itself (N = 240, ~760 M nonzeros); the default single-GPU stand-in
    uses scale = 1/64 (N = 60, ~22.7 M nonzeros)."""
    return syn_nlpkkt(max(3, int(round(240.0 * scale ** (1.0 / 3.0)))))


WORKLOADS = {
    "syn-cant": syn_cant,
    "syn-nd24k": syn_nd24k,
    "syn-webbase": syn_webbase,
    "syn-nlpkkt": syn_nlpkkt_scaled,
    "syn-bandrandom": lambda scale=1.0: syn_bandrandom(max(2000, int(200000 * scale))),
}
