"""spx_hip_mat_x_pieces: which pieces of x every row-block of the descriptor stream reads (stream_rowblock_xpieces).
The host-vector entry point sends x in the order the parts of the product need it (device_spmv_host): a piece that
the mask of a row-block misses would reach the device after the row-block has run.  Checked against CSR on host-only
tunes: for every row-block the pieces that hold the columns of its rows (symmetric streams: of the stored lower
triangle, plus the rows themselves -- the diagonal term and the transposed products read x there) are in its mask,
and the masks are not trivially full."""
import ctypes as C

import numpy as np
import pytest

import sparsex_amd as sx
from sparsex_amd import synth
from helpers import tune

CASES = [("kkt", lambda: synth.syn_nlpkkt(20), {}, False),
         ("kkt-plain", lambda: synth.syn_nlpkkt(18), {"spx.gpu.unit_windows": "false"}, False),
         ("cant", lambda: synth.syn_cant(0.3), {}, False),
         ("web", lambda: synth.syn_webbase(0.1), {}, False),
         ("kkt-sym", lambda: synth.syn_nlpkkt(16), {}, True),
         ("kkt-sym-segments", lambda: synth.syn_nlpkkt(30), {"spx.gpu.sym_segments": "true"}, True),
         ("nd24k-sym", lambda: synth.syn_nd24k(0.08), {}, True)]


def masks(A, piece):
    return A.x_pieces(piece)


@pytest.mark.parametrize("name,make,opts,sym", CASES, ids=[c[0] for c in CASES])
def test_masks_cover_what_csr_says(name, make, opts, sym):
    csr = make()
    rp, ci, n = np.asarray(csr[0], dtype=np.int64), np.asarray(csr[1], dtype=np.int64), csr[3]
    rp, ci = rp - rp[0], ci - int(np.asarray(csr[0])[0])
    A = tune(csr, opts, sym=sym, host_only=True)
    piece = max(512, (n + 63) // 64 + 7 & ~7)
    m, r0, nr = masks(A, piece)
    assert len(m) > 0
    rows_of = np.repeat(np.arange(n), np.diff(rp))
    # every nonzero of the matrix is covered by a row-block that holds its row (general streams; symmetric streams
    # hold the lower triangle, and what they hold of the upper one lies in the same rows)
    covered = np.zeros(len(ci), dtype=bool)
    for i in range(len(m)):
        lo, hi = int(r0[i]), int(r0[i]) + int(nr[i])
        a, b = rp[lo], rp[hi]
        c, r = ci[a:b], rows_of[a:b]
        if sym:
            keep = c < r
            c = c[keep]
            own = np.arange(lo, hi) // piece
            assert all((int(m[i]) >> int(p)) & 1 for p in np.unique(own)), (name, i, "own rows")
        need = np.unique(c // piece)
        have = int(m[i])
        if not sym:
            # (column slices: a row-block holds only its slice of the columns -- then the union over the
            # row-blocks of a row range must cover; checked below)
            covered[a:b] |= np.array([(have >> int(p)) & 1 for p in c // piece], dtype=bool) if len(c) else False
        else:
            missing = [int(p) for p in need if not (have >> int(p)) & 1]
            # (rows whose lower part lies in tiles or segments of another row-block do not exist: a row-block
            # holds all stored entries of its rows)
            assert not missing, (name, i, lo, hi, missing)
    if not sym:
        assert covered.all(), (name, int((~covered).sum()))
    # not trivially full: on these banded / block matrices most row-blocks read a few pieces
    bits = np.array([bin(int(v)).count("1") for v in m])
    if name != "web":
        assert np.median(bits) <= 16, (name, np.median(bits))
    A.destroy()
    sx.options_reset()


def test_more_than_64_pieces_is_an_error():
    A = tune(synth.syn_nlpkkt(12), {}, host_only=True)
    L = sx.lib()
    L.spx_hip_mat_x_pieces.restype = C.c_int64
    L.spx_hip_mat_x_pieces.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.spx_log_disable_all()
    assert L.spx_hip_mat_x_pieces(C.c_void_p(A.handle), 8, None, None, None, 0) == -1
    assert L.spx_hip_mat_x_pieces(C.c_void_p(A.handle), 0, None, None, None, 0) == -1
    L.spx_log_error_console()
    A.destroy()
    sx.options_reset()
