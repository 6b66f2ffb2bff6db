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
    """The nlpkkt stand-in of SURVEY.md section 8(d): KKT system [H A^T; A D] of order
    2 N^3 + 6 N^2 -- N^3 states, 6 N^2 boundary controls, N^3 multipliers; H and D
    diagonal, A = [27-point stencil of the grid | one entry per boundary face] -- about
    27 nonzeros per row in runs of three consecutive columns, symmetric, full diagonal
    (nlpkkt240 is N = 240: 27 993 600 rows, ~760 M nonzeros; tests use small N).  The
    pattern is that of tools/synth/nlpkkt_gen.c (syn_nlpkkt_rows)."""
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    N3, N2 = N ** 3, N ** 2
    P = N3 + 6 * N2
    n = P + N3
    idx = np.arange(N3)
    z, y, x = idx // N2, (idx // N) % N, idx % N
    rows, cols = [np.arange(n)], [np.arange(n)]
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                zz, yy, xx = z + dz, y + dy, x + dx
                ok = (zz >= 0) & (zz < N) & (yy >= 0) & (yy < N) & (xx >= 0) & (xx < N)
                rows.append(P + idx[ok])                       # A_y: multiplier row, state column
                cols.append((zz * N2 + yy * N + xx)[ok])
    a, b = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
    a, b = a.ravel(), b.ravel()
    cells = [a * N + b, (N - 1) * N2 + a * N + b, a * N2 + b, a * N2 + (N - 1) * N + b,
             a * N2 + b * N, a * N2 + b * N + (N - 1)]
    for q, cell in enumerate(cells):                           # A_u: one entry per face
        rows.append(P + cell)
        cols.append(N3 + q * N2 + np.arange(N2))
    r = np.concatenate(rows)
    c = np.concatenate(cols)
    r, c = np.concatenate([r, c]), np.concatenate([c, r])     # + A^T
    return _finish(r, c, n, rng, symmetric=True)


def syn_kkt2f(N=12, seed=SEED_BASE + 4):
    """Two fully coupled interleaved fields on an N^3 grid with 27-point stencils plus
    6 N^2 constraint rows: 54 nonzeros per row in runs of six consecutive columns.
    (Rounds 1-2 used this as their nlpkkt stand-in; it is twice as dense per row as
    nlpkkt240 and kept for comparison only.)"""
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
    """libspxsynth.so (tools/synth/nlpkkt_gen.c and kkt2f_gen.c, built by `make lib`):
    the row-sliced generators of the nlpkkt stand-in and of syn-kkt2f."""
    global _SYNLIB
    if _SYNLIB is None:
        import ctypes as C
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libspxsynth.so")
        L = C.CDLL(path)
        for g in ("nlpkkt", "kkt2f"):
            f = getattr(L, "spx_syn_%s_nrows" % g)
            f.restype, f.argtypes = C.c_int64, [C.c_int]
            f = getattr(L, "spx_syn_%s_counts" % g)
            f.restype, f.argtypes = None, [C.c_int, C.c_void_p]
            f = getattr(L, "spx_syn_%s_rows" % g)
            f.restype = C.c_int64
            f.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.spx_syn_nlpkkt_rows_perm.restype = C.c_int64
        L.spx_syn_nlpkkt_rows_perm.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_uint64, C.c_void_p,
                                               C.c_void_p, C.c_void_p]
        _SYNLIB = L
    return _SYNLIB


def _nrows(gen, N):
    return int(getattr(_synlib(), "spx_syn_%s_nrows" % gen)(int(N)))


def _row_counts(gen, N):
    cnt = np.empty(_nrows(gen, N), dtype=np.int32)
    getattr(_synlib(), "spx_syn_%s_counts" % gen)(int(N), cnt.ctypes.data)
    return cnt


def _rows(gen, N, lo, hi, counts, seed):
    n = _nrows(gen, N)
    hi = n if hi is None else hi
    if counts is None:
        counts = _row_counts(gen, N)
    rp = np.zeros(hi - lo + 1, dtype=np.int64)
    np.cumsum(counts[lo:hi], dtype=np.int64, out=rp[1:])
    nnz = int(rp[-1])
    assert nnz < 2 ** 31, "slice too large for 32-bit row pointers"
    ci = np.empty(nnz, dtype=np.int32)
    va = np.empty(nnz, dtype=np.float64)
    fn = getattr(_synlib(), "spx_syn_%s_rows" % gen)
    # large slices: row ranges of equal nonzero counts on a few threads (the C call drops the GIL)
    import os
    from concurrent.futures import ThreadPoolExecutor
    T = max(1, min(len(os.sched_getaffinity(0)), 32, nnz >> 22))
    cuts = [int(np.searchsorted(rp, nnz * t // T, side="left")) for t in range(T)] + [hi - lo]

    def piece(t):
        a, b = cuts[t], cuts[t + 1]
        if b <= a:
            return 0
        rpl = np.empty(b - a + 1, dtype=np.int64)
        k0 = int(rp[a])
        return fn(int(N), int(lo + a), int(lo + b), int(seed), rpl.ctypes.data,
                  ci.ctypes.data + 4 * k0, va.ctypes.data + 8 * k0)
    with ThreadPoolExecutor(T) as ex:
        got = sum(ex.map(piece, range(T)))
    assert got == nnz
    return rp.astype(np.int32), ci, va, n


def _pattern(gen, N, counts=None):
    """Row pointers and column indices of the whole matrix, no values (what a partition-aware
    numbering is computed from: sparsex_amd.dist_reorder)."""
    assert gen == "nlpkkt", "pattern-only generation exists for syn-nlpkkt"
    n = _nrows(gen, N)
    if counts is None:
        counts = _row_counts(gen, N)
    rp = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, dtype=np.int64, out=rp[1:])
    nnz = int(rp[-1])
    assert nnz < 2 ** 31
    ci = np.empty(nnz, dtype=np.int32)
    fn = getattr(_synlib(), "spx_syn_%s_rows" % gen)
    import os
    from concurrent.futures import ThreadPoolExecutor
    T = max(1, min(len(os.sched_getaffinity(0)), 32, nnz >> 22))
    cuts = [int(np.searchsorted(rp, nnz * t // T, side="left")) for t in range(T)] + [n]

    def piece(t):
        a, b = cuts[t], cuts[t + 1]
        if b <= a:
            return 0
        rpl = np.empty(b - a + 1, dtype=np.int64)
        return fn(int(N), int(a), int(b), 0, rpl.ctypes.data, ci.ctypes.data + 4 * int(rp[a]), None)
    with ThreadPoolExecutor(T) as ex:
        got = sum(ex.map(piece, range(T)))
    assert got == nnz
    return rp.astype(np.int32), ci


def _rows_perm(gen, N, rows_old, perm, counts, seed):
    """The rows `rows_old` (original numbering) of P A P^T, perm[old] = new: CSR with renumbered,
    sorted columns (tools/synth/nlpkkt_gen.c::spx_syn_nlpkkt_rows_perm)."""
    assert gen == "nlpkkt", "permuted row generation exists for syn-nlpkkt"
    rows_old = np.ascontiguousarray(rows_old, dtype=np.int64)
    perm = np.ascontiguousarray(perm, dtype=np.int32)
    n = _nrows(gen, N)
    m = rows_old.size
    rp = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(counts[rows_old], dtype=np.int64, out=rp[1:])
    nnz = int(rp[-1])
    assert nnz < 2 ** 31, "slice too large for 32-bit row pointers"
    ci = np.empty(nnz, dtype=np.int32)
    va = np.empty(nnz, dtype=np.float64)
    fn = _synlib().spx_syn_nlpkkt_rows_perm
    import os
    from concurrent.futures import ThreadPoolExecutor
    T = max(1, min(len(os.sched_getaffinity(0)), 32, nnz >> 22))
    cuts = [int(np.searchsorted(rp, nnz * t // T, side="left")) for t in range(T)] + [m]

    def piece(t):
        a, b = cuts[t], cuts[t + 1]
        if b <= a:
            return 0
        rpl = np.empty(b - a + 1, dtype=np.int64)
        k0 = int(rp[a])
        return fn(int(N), rows_old.ctypes.data + 8 * a, int(b - a), perm.ctypes.data, int(seed), rpl.ctypes.data,
                  ci.ctypes.data + 4 * k0, va.ctypes.data + 8 * k0)
    with ThreadPoolExecutor(T) as ex:
        got = sum(ex.map(piece, range(T)))
    assert got == nnz
    return rp.astype(np.int32), ci, va, n


def nlpkkt_nrows(N):
    return _nrows("nlpkkt", N)


def nlpkkt_row_counts(N):
    """Nonzeros of every row of syn_nlpkkt_rows(N) (int32, all rows)."""
    return _row_counts("nlpkkt", N)


def syn_nlpkkt_rows(N, lo=0, hi=None, counts=None, seed=SEED_BASE + 4):
    """Rows [lo, hi) of the nlpkkt stand-in as a CSR slice: the pattern of
    syn_nlpkkt(N), generated row by row in C so that a process can hold only
    the rows it owns (values: symmetric hash of the coordinate pair, diagonal
    dominant).  Returns (rowptr int32 relative to the slice, colind int32,
    values float64, n) with n the order of the WHOLE matrix."""
    return _rows("nlpkkt", N, lo, hi, counts, seed)


def kkt2f_nrows(N):
    return _nrows("kkt2f", N)


def kkt2f_row_counts(N):
    return _row_counts("kkt2f", N)


def syn_kkt2f_rows(N, lo=0, hi=None, counts=None, seed=SEED_BASE + 4):
    """Rows [lo, hi) of syn-kkt2f (the pattern of syn_kkt2f(N)), generated in C."""
    return _rows("kkt2f", N, lo, hi, counts, seed)


def stored_row_counts(gen, N, counts=None):
    """Per row: the nonzeros on and below the diagonal -- what the symmetric path stores, and
    what its rows are balanced by when they are dealt to several processes."""
    if counts is None:
        counts = _row_counts(gen, N)
    if gen == "nlpkkt":
        # states and controls keep their diagonal, the multiplier rows everything
        # (tools/synth/nlpkkt_gen.c: the blocks A_y, A_u lie below the diagonal)
        P = int(N) ** 3 + 6 * int(N) ** 2
        out = counts.copy()
        out[:P] = 1
        return out
    out = np.empty_like(counts)
    n = counts.size
    step = 1 << 20
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        rp, ci, _, _ = _rows(gen, N, lo, hi, counts, SEED_BASE + 4)
        rows = np.repeat(np.arange(lo, hi, dtype=np.int64), np.diff(rp))
        out[lo:hi] = np.bincount((rows - lo)[ci <= rows], minlength=hi - lo)
    return out


def nlpkkt_edge(scale):
    """Grid edge for a size factor: scale 1 is nlpkkt240's order (N = 240)."""
    return max(3, int(round(240.0 * scale ** (1.0 / 3.0))))


def syn_nlpkkt_scaled(scale=1.0):
    """syn_nlpkkt_rows with the grid edge derived from a size factor: scale 1 is the
    order of nlpkkt240 itself (N = 240, 769 M nonzeros)."""
    return syn_nlpkkt_rows(nlpkkt_edge(scale))


WORKLOADS = {
    "syn-cant": syn_cant,
    "syn-nd24k": syn_nd24k,
    "syn-webbase": syn_webbase,
    "syn-nlpkkt": syn_nlpkkt_scaled,
    "syn-kkt2f": lambda scale=1.0: syn_kkt2f_rows(nlpkkt_edge(scale)),
    "syn-bandrandom": lambda scale=1.0: syn_bandrandom(max(2000, int(200000 * scale))),
}
