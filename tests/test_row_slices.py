"""Row slices: a process hands spx_input_load_csr only the rows it owns of a larger
matrix (spx.rt.row_offset / spx.rt.global_rows) -- the multi-GPU layout of
BASELINE config 4, where no rank can afford the whole nlpkkt240.  Checked on the
CPU through the independent numpy decoder of the saved streams: general slices
tile the matrix, symmetric slices give partial vectors that sum to A x."""
import numpy as np
import pytest
import scipy.sparse as sp

import sparsex_amd as sx
from sparsex_amd import synth
from helpers import tune
from stream_decode import Stream


def nnz_balanced_bounds(counts, world):
    cum = np.concatenate([[0], np.cumsum(counts, dtype=np.int64)])
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(cum, cum[-1] * r // world)))
    return cuts + [counts.size]


def slice_opts(lo, n, extra=None):
    o = {"spx.rt.row_offset": str(lo), "spx.rt.global_rows": str(n), "spx.preproc.sampling": "none",
         "spx.rt.nr_threads": "2"}
    o.update(extra or {})
    return o


def test_generator_slices_are_rows_of_the_whole():
    N = 9
    rp, ci, va, n = synth.syn_nlpkkt_rows(N)
    rp0, ci0, _, n0 = synth.syn_nlpkkt(N)                 # the scipy generator: same pattern
    assert n == n0 and np.array_equal(rp, rp0) and np.array_equal(ci, ci0)
    a = sp.csr_matrix((va, ci, rp), shape=(n, n))
    assert abs(a - a.T).max() == 0 and (a.diagonal() > 0).all()
    cnt = synth.nlpkkt_row_counts(N)
    assert np.array_equal(cnt, np.diff(rp))
    for lo, hi in ((0, 7), (n // 3, n // 2), (n - 5, n)):
        r2, c2, v2, _ = synth.syn_nlpkkt_rows(N, lo, hi)
        assert np.array_equal(c2, ci[rp[lo]:rp[hi]]) and np.array_equal(v2, va[rp[lo]:rp[hi]])
        assert np.array_equal(r2, rp[lo:hi + 1] - rp[lo])


@pytest.mark.parametrize("world", [2, 3])
def test_general_slices_tile_the_matrix(tmp_path, world):
    N = 8
    rp, ci, va, n = synth.syn_nlpkkt_rows(N)
    a = sp.csr_matrix((va, ci, rp), shape=(n, n))
    cuts = nnz_balanced_bounds(synth.nlpkkt_row_counts(N), world)
    x = synth.random_x(n)
    y = np.zeros(n)
    for r in range(world):
        lo, hi = cuts[r], cuts[r + 1]
        rl, cl, vl, _ = synth.syn_nlpkkt_rows(N, lo, hi)
        sx.options_reset()
        for k, v in slice_opts(lo, n, {"spx.rt.host_only": "true"}).items():
            sx.option_set(k, v)
        A = sx.mat_tune(sx.input_load_csr(rl, cl, vl, hi - lo, n))
        inf = A.info()
        assert (inf.row_lo, inf.row_hi) == (lo, hi) and A.nrows == n
        f = str(tmp_path / ("g%d.spx" % r))
        A.save(f)
        s = Stream(f)
        rr, cc, vv, _ = s.triplets()
        assert rr.min() >= lo and rr.max() < hi               # only its own rows
        m = sp.coo_matrix((vv, (rr, cc)), shape=(n, n)).tocsr()
        assert abs(m[lo:hi] - a[lo:hi]).max() == 0
        # entries are addressed globally
        j = rp[lo] + 3
        assert A.get_entry(lo, int(ci[j - 3])) == va[j - 3]
        y += s.matvec(x)
    assert np.allclose(y, a @ x, rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("gen", ["nlpkkt", "kkt2f", "nd24k"])
@pytest.mark.parametrize("world", [2, 3])
def test_symmetric_slices_sum_to_the_product(tmp_path, world, gen):
    if gen == "nlpkkt":
        rp, ci, va, n = synth.syn_nlpkkt_rows(8)
    elif gen == "kkt2f":
        rp, ci, va, n = synth.syn_kkt2f_rows(8)
    else:
        rp, ci, va, n = synth.syn_nd24k(0.02)
    a = sp.csr_matrix((va, ci, rp), shape=(n, n))
    cuts = nnz_balanced_bounds(np.diff(rp), world)
    x = synth.random_x(n)
    y = np.zeros(n)
    tiles = 0
    for r in range(world):
        lo, hi = cuts[r], cuts[r + 1]
        rl = (rp[lo:hi + 1] - rp[lo]).astype(np.int32)
        cl, vl = ci[rp[lo]:rp[hi]].copy(), va[rp[lo]:rp[hi]].copy()
        sx.options_reset()
        for k, v in slice_opts(lo, n, {"spx.rt.host_only": "true", "spx.matrix.symmetric": "true"}).items():
            sx.option_set(k, v)
        A = sx.mat_tune(sx.input_load_csr(rl, cl, vl, hi - lo, n))
        inf = A.info()
        assert (inf.row_lo, inf.row_hi) == (lo, hi)
        f = str(tmp_path / ("s%d.spx" % r))
        A.save(f)
        s = Stream(f)
        assert s.sym_fused == (world == 1)
        rr, cc, vv, _ = s.triplets()
        assert rr.size == 0 or rr.max() < hi                                   # nothing below its own rows
        tiles += int((s.passes["kind"] == 3).sum())
        y += s.matvec(x)
        # the diagonal and an off-diagonal entry of an owned row, global numbering
        assert A.get_entry(lo, lo) == a[lo, lo]
    assert np.allclose(y, a @ x, rtol=1e-12, atol=1e-14)
    assert (tiles > 0) == (gen == "nd24k")


def test_thin_mirror_image_is_kept_per_row(tmp_path):
    """The last slice of syn-kkt2f couples, through its constraint rows, into rows all over
    the grid: a handful of mirrored nonzeros per 512 rows.  They do not become row-blocks of
    their own (a workgroup per handful of nonzeros doubled that rank's SpMV time) but a small
    per-row list; entries in it can be read and set like any other."""
    N, world = 40, 3
    rp, ci, va, n = synth.syn_kkt2f_rows(N)
    va = va.copy()
    cuts = nnz_balanced_bounds(np.diff(rp), world)
    lo, hi = cuts[-2], cuts[-1]
    rl = (rp[lo:hi + 1] - rp[lo]).astype(np.int32)
    cl, vl = ci[rp[lo]:rp[hi]].copy(), va[rp[lo]:rp[hi]].copy()
    sx.options_reset()
    o = slice_opts(lo, n, {"spx.rt.host_only": "true", "spx.matrix.symmetric": "true"})
    del o["spx.preproc.sampling"]
    for k, v in o.items():
        sx.option_set(k, v)
    A = sx.mat_tune(sx.input_load_csr(rl, cl, vl, hi - lo, n))
    f = str(tmp_path / "last.spx")
    A.save(f)
    s = Stream(f)
    assert s.mirror_rows.size > 1000 and s.mirror_rows.max() < lo
    # the row-blocks in front of the slice cover the band only, the list took the thin part
    # (none at all where the band went into read-once row segments, which add into y themselves)
    front = s.rbs[s.rbs["row0"] < lo]
    assert front.size == 0 or int(front["row0"].min()) > lo - 3 * (N * N + N + 1)
    # an entry whose mirror image lives in the list: (constraint row, coupled grid row)
    listed = set(s.mirror_rows.tolist())
    r = next(q for q in range(n - 1, lo, -1) if int(ci[rp[q]]) in listed)
    k = int(rp[r])
    c = int(ci[k])                          # its first column: a grid row far in front of the slice
    assert c < lo
    assert A.get_entry(r, c) == va[k] and A.get_entry(c, r) == va[k]
    A.set_entry(r, c, 4.5)
    assert A.get_entry(c, r) == 4.5
    A.save(f)
    s2 = Stream(f)
    va[k] = 4.5
    va[rp[c] + int(np.searchsorted(ci[rp[c]:rp[c + 1]], r))] = 4.5
    full = sp.csr_matrix((va, ci, rp), shape=(n, n))
    x = synth.random_x(n)
    lower = sp.tril(full, k=-1).tocsr()[lo:hi]               # strictly lower part of the slice's rows
    want = np.zeros(n)
    want[lo:hi] = lower @ x + full.diagonal()[lo:hi] * x[lo:hi]
    want += lower.T @ x[lo:hi]
    assert np.allclose(s2.matvec(x), want, rtol=1e-12, atol=1e-14)
