"""The HBM layout on the CPU: an independent numpy decoder (tests/stream_decode.py)
reads the saved row-block descriptor stream lane by lane and must reproduce the
input matrix exactly -- every nonzero once, in the row-block that owns its row."""
import numpy as np
import pytest
import scipy.sparse as sp

import sparsex_amd as sx
from sparsex_amd import synth
from helpers import tune
from stream_decode import Stream

def zoo(n=1500, seed=9):
    """Vertical / diagonal / anti-diagonal lines with strides 1..4, strided horizontal runs, blocks."""
    rng = np.random.RandomState(seed)
    rows, cols = [np.arange(n)], [np.arange(n)]
    for k in range(60):
        r0, c0 = rng.randint(0, n - 400), rng.randint(200, n - 400)
        ln, st = rng.randint(5, 90), 1 + k % 4
        t = np.arange(ln) * st
        shape = k % 5
        if shape == 0: rows.append(r0 + t); cols.append(np.full(ln, c0))            # vertical
        elif shape == 1: rows.append(r0 + t); cols.append(c0 + t)                   # diagonal
        elif shape == 2: rows.append(r0 + t); cols.append(c0 + 380 - t)             # anti-diagonal
        elif shape == 3: rows.append(np.full(ln, r0)); cols.append(c0 + t)          # horizontal
        else:                                                                        # dense block
            a, b = np.meshgrid(np.arange(rng.randint(2, 9)), np.arange(rng.randint(2, 20)), indexing="ij")
            rows.append(r0 + a.ravel()); cols.append(c0 + b.ravel())
    m = sp.coo_matrix((np.ones(sum(x.size for x in rows)), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(n, n)).tocsr()
    m.sum_duplicates(); m.sort_indices()
    m.data = rng.uniform(0.5, 1.5, m.nnz)
    return (m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.copy(), n)


CASES = [
    ("zoo-all", zoo, {"spx.preproc.xform": "all", "spx.preproc.sampling": "none"}),
    ("zoo-v", zoo, {"spx.preproc.xform": "v", "spx.preproc.sampling": "none", "spx.gpu.rowblock_rows": "37"}),
    ("zoo-ad", zoo, {"spx.preproc.xform": "ad", "spx.preproc.sampling": "none", "spx.gpu.rowblock_rows": "50"}),
    ("zoo-d", zoo, {"spx.preproc.xform": "d", "spx.preproc.sampling": "none", "spx.gpu.rowblock_rows": "19"}),
    ("zoo-strided", zoo, {"spx.preproc.xform": "v{2},ad{3},d{2},h{4}", "spx.gpu.rowblock_rows": "23"}),
    ("zoo-h", zoo, {"spx.preproc.xform": "h", "spx.preproc.sampling": "none"}),
    ("zoo-br", zoo, {"spx.preproc.xform": "br", "spx.preproc.sampling": "none", "spx.gpu.rowblock_rows": "3"}),
    ("cant", lambda: synth.syn_cant(0.04), {"spx.preproc.sampling": "none"}),
    ("cant-p3", lambda: synth.syn_cant(0.04), {"spx.preproc.sampling": "none", "spx.rt.nr_threads": "3"}),
    ("cant-small-rb", lambda: synth.syn_cant(0.02), {"spx.preproc.sampling": "none", "spx.gpu.rowblock_elems": "300",
                                                     "spx.gpu.rowblock_rows": "7"}),
    ("nd24k", lambda: synth.syn_nd24k(0.02), {"spx.preproc.sampling": "none"}),
    ("nd24k-nostack", lambda: synth.syn_nd24k(0.02), {"spx.preproc.sampling": "none", "spx.gpu.stack_segments": "false"}),
    ("webbase", lambda: synth.syn_webbase(0.01), {}),
    ("webbase-wide", lambda: synth.syn_webbase(0.08), {}),            # u32 column offsets
    ("webbase-tiny-rb", lambda: synth.syn_webbase(0.01), {"spx.gpu.rowblock_elems": "100", "spx.gpu.rowblock_rows": "11"}),
    ("nlpkkt", lambda: synth.syn_nlpkkt(7), {"spx.preproc.sampling": "none", "spx.rt.nr_threads": "2"}),
    ("all-types", lambda: synth.syn_nlpkkt(6), {"spx.preproc.xform": "all", "spx.preproc.sampling": "none"}),
    # enough row-blocks for the emitter to build runs of them on several threads and join the pieces
    ("cant-threaded", lambda: synth.syn_cant(0.25), {"spx.gpu.rowblock_elems": "400"}),
    ("webbase-threaded", lambda: synth.syn_webbase(0.1), {"spx.gpu.rowblock_elems": "300", "spx.rt.nr_threads": "3"}),
    # leftovers inside a narrow band: row-blocks stage an x window in LDS, u16 offsets from its base
    ("band-window", lambda: synth.syn_bandrandom(8000), {}),
    ("band-no-window", lambda: synth.syn_bandrandom(8000), {"spx.gpu.x_window": "false"}),
    # few nonzeros per row: row-blocks of up to 2048 rows, joined from planned ones of <= 512
    ("web-wide-rows", lambda: synth.syn_webbase(0.1), {"spx.gpu.rowblock_rows": "2048", "spx.gpu.rowblock_elems": "6000",
                                                        "spx.rt.nr_threads": "2"}),
]


def dense_of(stream):
    r, c, v, b = stream.triplets()
    m = sp.coo_matrix((v, (r, c)), shape=(stream.nrows, stream.ncols))
    return r, c, v, b, m


@pytest.mark.parametrize("name,gen,opts", CASES, ids=[c[0] for c in CASES])
def test_general_stream_holds_the_matrix_exactly(tmp_path, name, gen, opts):
    csr = gen()
    rp, ci, va, n = csr
    A = tune(csr, opts, host_only=True)
    f = str(tmp_path / "m.spx")
    A.save(f)
    s = Stream(f)
    r, c, v, b, m = dense_of(s)
    assert r.size == rp[-1] == s.nnz_stored               # every nonzero exactly once
    a = sp.csr_matrix((va, ci, rp), shape=(n, n))
    assert abs(m.tocsr() - a).max() == 0                  # values bit-identical, right places
    assert np.unique(r * n + c).size == r.size            # no duplicates hiding behind a sum
    s.check_ownership()
    row0 = s.rbs["row0"].astype(np.int64)[b]
    assert ((r >= row0) & (r < row0 + s.rbs["n_rows"].astype(np.int64)[b])).all()
    if name == "web-wide-rows":
        assert int(s.rbs["n_rows"].max()) > 1024
    if name.startswith("band"):
        windows = int((s.rbs["xwin_len"] > 0).sum())
        assert (windows > len(s.rbs) // 2) == (name == "band-window")
        assert bool((s.passes["kind"] == 4).any()) == (name == "band-window")


@pytest.mark.parametrize("remine", ["true", "false"])
@pytest.mark.parametrize("name,gen,opts", [
    ("cant", lambda: synth.syn_cant(0.04), {"spx.preproc.sampling": "none", "spx.rt.nr_threads": "2"}),
    ("nd24k", lambda: synth.syn_nd24k(0.02), {"spx.preproc.sampling": "none"}),
    ("nlpkkt-all", lambda: synth.syn_nlpkkt(6), {"spx.preproc.xform": "all", "spx.preproc.sampling": "none"}),
    ("nd24k-threaded", lambda: synth.syn_nd24k(0.08), {"spx.gpu.rowblock_elems": "1500", "spx.rt.nr_threads": "3"}),
    ("nlpkkt-threaded", lambda: synth.syn_nlpkkt(14), {"spx.gpu.rowblock_elems": "600", "spx.rt.nr_threads": "2"}),
], ids=["cant", "nd24k", "nlpkkt-all", "nd24k-threaded", "nlpkkt-threaded"])
def test_symmetric_stream_is_lower_plus_mirror(tmp_path, name, gen, opts, remine):
    csr = gen()
    rp, ci, va, n = csr
    o = dict(opts)
    o["spx.gpu.sym_remine"] = remine
    A = tune(csr, o, sym=True, host_only=True)
    f = str(tmp_path / "m.spx")
    A.save(f)
    s = Stream(f)
    r, c, v, b, m = dense_of(s)
    a = sp.csr_matrix((va, ci, rp), shape=(n, n))
    off = a - sp.diags(a.diagonal())
    assert abs(m.tocsr() - off.tocsr()).max() == 0        # strict lower + strict upper
    assert np.array_equal(s.dvalues, a.diagonal())
    owner = s.check_ownership()
    # the whole matrix in one process: every row has exactly one owner, whose
    # write-out adds the diagonal term (no separate init pass)
    assert s.sym_fused and owner[:n].sum() + len(s.shared) == n
    x = synth.random_x(n)
    assert np.allclose(s.matvec(x), a @ x, rtol=1e-12, atol=1e-14)


def test_zoo_cases_cover_every_descriptor_kind(tmp_path):
    seen = set()
    for xf in ("v", "ad", "d", "h", "br", "v{2},ad{3},d{2},h{4}"):
        A = tune(zoo(), {"spx.preproc.xform": xf, "spx.preproc.sampling": "none"}, host_only=True)
        f = str(tmp_path / "m.spx")
        A.save(f)
        s = Stream(f)
        bits = s.descs["bits"].astype(np.int64)
        seen |= set(zip(((bits >> 22) & 7).tolist(), (bits >> 25).tolist()))
    kinds = {k for k, _ in seen}
    assert kinds == {0, 1, 2, 3, 4}, kinds
    assert any(st > 1 for k, st in seen if k == 2) and any(st > 1 for k, st in seen if k == 4)


def test_long_rows_are_shared_and_complete(tmp_path):
    n = 30000
    rng = np.random.RandomState(5)
    rows = np.concatenate([np.full(20000, 7), np.full(9000, 11), np.arange(n)])
    cols = np.concatenate([rng.choice(n, 20000, replace=False), rng.choice(n, 9000, replace=False), np.arange(n)])
    a = sp.coo_matrix((rng.uniform(0.5, 1.5, rows.size), (rows, cols)), shape=(n, n)).tocsr()
    a.sum_duplicates(); a.sort_indices()
    csr = (a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data.copy(), n)
    A = tune(csr, {}, host_only=True)
    f = str(tmp_path / "m.spx")
    A.save(f)
    s = Stream(f)
    assert len(s.shared) == 2 and s.n_carry == sum(int(x["n_slots"]) for x in s.shared)
    r, c, v, b, m = dense_of(s)
    assert abs(m.tocsr() - a).max() == 0
    s.check_ownership()


def long_row_sym(n=30000, seed=8):
    """Symmetric, with two rows near the end whose lower parts (20000 and 9000 nonzeros, long dense
    runs among them) exceed a row-block, on top of a band with runs of four."""
    rng = np.random.RandomState(seed)
    r1 = np.full(20000, n - 7); c1 = np.arange(100, 20100)                       # one dense run
    r2 = np.full(9000, n - 3); c2 = np.sort(rng.choice(n - 10, 9000, replace=False))
    rb = np.repeat(np.arange(8, n), 4); cb = rb - np.tile(np.arange(5, 1, -1), n - 8)
    r, c = np.concatenate([r1, r2, rb]), np.concatenate([c1, c2, cb])
    low = sp.coo_matrix((rng.uniform(0.5, 1.5, r.size), (r, c)), shape=(n, n)).tocsr()
    low.sum_duplicates()
    low = sp.tril(low, k=-1)
    m = (low + low.T + sp.diags(rng.uniform(1.0, 2.0, n))).tocsr()
    m.sort_indices()
    return (m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.copy(), n), m


@pytest.mark.parametrize("segs", ["true", "false"])
def test_symmetric_long_rows_stay_on_the_mirrored_path(tmp_path, segs):
    """Rows too long for one row-block are chunked over several (shared rows); read-once segments
    leave them alone, whatever runs they hold."""
    csr, m = long_row_sym()
    rp, ci, va, n = csr
    A = tune(csr, {"spx.gpu.sym_segments": segs, "spx.rt.nr_threads": "3"}, sym=True, host_only=True)
    f = str(tmp_path / "m.spx")
    A.save(f)
    s = Stream(f)
    assert len(s.shared) >= 2
    assert (A.info().sym_segments > 0) == (segs == "true")
    r, c, v, b, got = dense_of(s)
    off = (m - sp.diags(m.diagonal())).tocsr()
    assert abs(got.tocsr() - off).max() == 0
    s.check_ownership()
    x = synth.random_x(n)
    assert np.allclose(s.matvec(x), m @ x, rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("once", ["true", "false"])
def test_symmetric_slices_sum_to_the_product(tmp_path, once):
    """Two processes, each holding half of the partitions of a symmetric matrix: their streams
    (tiles read once where there are any, the rest mirrored) give partial vectors that sum to A x."""
    csr = synth.syn_nd24k(0.03)
    rp, ci, va, n = csr
    a = sp.csr_matrix((va, ci, rp), shape=(n, n))
    x = synth.random_x(n)
    y = np.zeros(n)
    tiles = 0
    for rank in range(2):
        A = tune(csr, {"spx.preproc.sampling": "none", "spx.rt.nr_threads": "4", "spx.rt.gpu_world": "2",
                       "spx.rt.gpu_rank": str(rank), "spx.gpu.sym_once": once}, sym=True, host_only=True)
        f = str(tmp_path / ("m%d.spx" % rank))
        A.save(f)
        s = Stream(f)
        assert not s.sym_fused
        y += s.matvec(x)
        tiles += int((s.passes["kind"] == 3).sum())
    assert np.allclose(y, a @ x, rtol=1e-12, atol=1e-14)
    assert (tiles > 0) == (once == "true")


def _scattered_sym(n=6000, per_row=9, seed=23):
    """Random symmetric pattern with a few planted runs of 3..12 consecutive columns."""
    rng = np.random.RandomState(seed)
    r = np.repeat(np.arange(n), per_row)
    c = rng.randint(0, n, r.size)
    rr = rng.randint(20, n, 300)
    ln = rng.randint(3, 13, 300)
    c0 = (rng.uniform(0, 1, 300) * (rr - ln)).astype(np.int64)
    r = np.concatenate([r] + [np.full(l, q) for q, l in zip(rr, ln)])
    c = np.concatenate([c] + [np.arange(a, a + l) for a, l in zip(c0, ln)])
    keep = c < r
    low = sp.coo_matrix((rng.uniform(-1, 1, int(keep.sum())), (r[keep], c[keep])), shape=(n, n)).tocsr()
    low.sum_duplicates()
    m = (low + low.T + sp.diags(np.asarray(abs(low + low.T).sum(axis=1)).ravel() + 1.0)).tocsr()
    m.sort_indices()
    return m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.copy(), n


@pytest.mark.parametrize("threads", ["1", "3"])
@pytest.mark.parametrize("mode", ["true", "false", "auto"])
@pytest.mark.parametrize("name,gen,auto_on", [
    # (auto: only beyond 16 M nonzeros in the triangle -- tests/test_gpu_fullsize.py has such a case)
    ("kkt2f", lambda: synth.syn_kkt2f_rows(10), False),      # stencil rows: runs of four and six columns
    ("nlpkkt", lambda: synth.syn_nlpkkt_rows(10), False),    # KKT blocks: the triangle is the multiplier rows, runs of three
    ("cant", lambda: synth.syn_cant(0.05), False),
    ("nd24k", lambda: synth.syn_nd24k(0.02), False),          # dense tiles take the triangle first
    ("scattered", lambda: _scattered_sym(), False),           # hardly any runs
], ids=["kkt2f", "nlpkkt", "cant", "nd24k", "scattered"])
def test_symmetric_row_segments_are_stored_once(tmp_path, name, gen, auto_on, mode, threads):
    """spx.gpu.sym_segments: runs of >= 3 consecutive columns of the lower triangle become
    SPX_PASS_SYMSEG passes -- stored once, the lane forms the row sum AND hands value * x[row]
    to the columns' rows (through a slot of the row-block or straight into y)."""
    csr = gen()
    rp, ci, va, n = csr
    A = tune(csr, {"spx.gpu.sym_segments": mode, "spx.rt.nr_threads": threads, "spx.preproc.sampling": "none"},
             sym=True, host_only=True)
    f = str(tmp_path / "m.spx")
    A.save(f)
    s = Stream(f)
    used = np.zeros(len(s.passes), bool)
    for rb in s.rbs:
        used[int(rb["pass_off"]):int(rb["pass_off"]) + int(rb["n_pass"])] = True
    has = bool((s.passes["kind"][used] == 5).any())
    assert (A.info().sym_segments > 0) == has
    assert A.info().sym_segments != 1 or bool((s.passes["kind"][used] == 3).any())
    if mode == "false":
        assert not has
    elif mode == "true":
        assert has
    elif auto_on is not None:
        assert has == auto_on
    if has:
        assert s.sym_atomic and A.info().sym_tiles == 2
        # what was a mirrored copy is gone: about the lower triangle is stored, not twice that
        lower = synth.lower_plus_diag_nnz(rp, ci) - n
        assert s.nnz_stored < (2.0 if name == "scattered" else 1.35) * lower + n
    r, c, v, b, m = dense_of(s)
    a = sp.csr_matrix((va, ci, rp), shape=(n, n))
    off = a - sp.diags(a.diagonal())
    assert abs(m.tocsr() - off.tocsr()).max() == 0
    s.check_ownership()
    x = synth.random_x(n)
    assert np.allclose(s.matvec(x), a @ x, rtol=1e-12, atol=1e-14)
    # a restored matrix keeps them, and its entries can still be read and set from either triangle
    B = sx.mat_restore(f)
    assert B.info().sym_segments == A.info().sym_segments
    k = int(rp[n // 2]) if ci[rp[n // 2]] < n // 2 else int(rp[n // 2 + 1])
    row = int(np.searchsorted(rp, k, side="right") - 1)
    col = int(ci[k])
    if col < row:
        assert B.get_entry(row, col) == va[k] == B.get_entry(col, row)
        B.set_entry(col, row, 2.5)
        assert B.get_entry(row, col) == 2.5


@pytest.mark.parametrize("min_run", ["2", "3", "5"])
def test_shortest_read_once_run(tmp_path, min_run):
    """spx.gpu.sym_segment_min: shorter runs of the lower triangle stay on the mirrored path."""
    csr = synth.syn_kkt2f_rows(12)
    rp, ci, va, n = csr
    A = tune(csr, {"spx.gpu.sym_segments": "true", "spx.gpu.sym_segment_min": min_run, "spx.rt.nr_threads": "2"},
             sym=True, host_only=True)
    f = str(tmp_path / "m.spx")
    A.save(f)
    s = Stream(f)
    used = np.zeros(len(s.passes), bool)
    for rb in s.rbs:
        used[int(rb["pass_off"]):int(rb["pass_off"]) + int(rb["n_pass"])] = True
    seg = s.passes[used & (s.passes["kind"] == 5)]
    assert len(seg) and int(seg["width"].min()) >= int(min_run) and (int(seg["width"].min()) == 2) == (min_run == "2")
    a = sp.csr_matrix((va, ci, rp), shape=(n, n))
    x = synth.random_x(n)
    assert np.allclose(s.matvec(x), a @ x, rtol=1e-12, atol=1e-14)
    r, c, v, b, m = dense_of(s)
    assert abs(m.tocsr() - (a - sp.diags(a.diagonal())).tocsr()).max() == 0


def test_slot_groups_stay_inside_the_matrix(tmp_path):
    """Slots come in aligned groups of eight columns that are handed to y whole.  In front of a
    row-block of the last rows of a matrix whose order is not a multiple of eight, such a group
    would reach past the last row (found by tools/soak_random.py: heap corruption at tune time,
    out-of-bounds atomics at run time): those columns get no slots, their segments add to y
    themselves."""
    n = 700 + 3
    r = np.repeat(np.arange(6, n), 5)
    c = r - np.tile(np.arange(6, 1, -1), n - 6)                  # every row: a run of five columns in front of it
    rng = np.random.RandomState(4)
    low = sp.coo_matrix((rng.uniform(0.5, 1.5, r.size), (r, c)), shape=(n, n)).tocsr()
    m = (low + low.T + sp.diags(rng.uniform(1.0, 2.0, n))).tocsr()
    m.sort_indices()
    csr = (m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.copy(), n)
    for rows in ("3", "5", "512"):
        A = tune(csr, {"spx.gpu.sym_segments": "true", "spx.gpu.rowblock_rows": rows, "spx.rt.nr_threads": "2",
                       "spx.gpu.sym_wide_rows": "700"}, sym=True, host_only=True)
        f = str(tmp_path / "m.spx")
        A.save(f)
        s = Stream(f)
        assert s.slot_group_col.size == 0 or int(s.slot_group_col.max()) + 8 <= n
        x = synth.random_x(n)
        assert np.allclose(s.matvec(x), m @ x, rtol=1e-12, atol=1e-14)
        sx.mat_restore(f).destroy()                             # (the restore-time validation agrees)


@pytest.mark.parametrize("wide", ["512", "1024", "2048"])
def test_wide_rowblocks_share_their_slots(tmp_path, wide):
    """spx.gpu.sym_wide_rows: consecutive row-blocks with read-once segments go side by side into
    one (one y tile, one set of slots); their passes carry the first row of their part."""
    csr = synth.syn_kkt2f_rows(24)
    rp, ci, va, n = csr
    va = va.copy()
    A = tune((rp, ci, va, n), {"spx.gpu.sym_segments": "true", "spx.gpu.sym_wide_rows": wide, "spx.rt.nr_threads": "2"},
             sym=True, host_only=True)
    f = str(tmp_path / "m.spx")
    A.save(f)
    s = Stream(f)
    rows = s.rbs["n_rows"].astype(np.int64)
    assert rows.max() <= int(wide) and (rows.max() > 512) == (wide != "512")
    if wide != "512":
        # fewer slots in total: a column that two neighbouring row-blocks reached is one slot now
        B = tune((rp, ci, va, n), {"spx.gpu.sym_segments": "true", "spx.gpu.sym_wide_rows": "512", "spx.rt.nr_threads": "2"},
                 sym=True, host_only=True)
        g = str(tmp_path / "narrow.spx")
        B.save(g)
        assert int(s.rbs["n_slots"].astype(np.int64).sum()) < 0.8 * int(Stream(g).rbs["n_slots"].astype(np.int64).sum())
        wide_rb = s.rbs[rows > 512][0]
        bases = {int(s.passes[int(wide_rb["pass_off"]) + t]["elem0"]) for t in range(int(wide_rb["n_pass"]))
                 if int(s.passes[int(wide_rb["pass_off"]) + t]["kind"]) in (0, 5)}
        assert len(bases) >= 2 and 0 in bases and max(bases) < int(wide_rb["n_rows"])
    a = sp.csr_matrix((va, ci, rp), shape=(n, n))
    r, c, v, b, m = dense_of(s)
    off = a - sp.diags(a.diagonal())
    assert abs(m.tocsr() - off.tocsr()).max() == 0
    s.check_ownership()
    x = synth.random_x(n)
    assert np.allclose(s.matvec(x), a @ x, rtol=1e-12, atol=1e-14)
    # entries of a part that is not the first of its row-block: found, set, and kept by a restore
    big = s.rbs[np.argmax(rows)]
    row = int(big["row0"]) + int(big["n_rows"]) - 3
    k = int(rp[row])
    col = int(ci[k])
    assert col < row and A.get_entry(row, col) == va[k] == A.get_entry(col, row)
    A.set_entry(row, col, -7.25)
    A.save(f)
    B = sx.mat_restore(f)
    assert B.get_entry(col, row) == -7.25 and B.info().n_rowblocks == len(s.rbs)


def _one_rowblock_sym():
    """12 x 12 blocks of 8 x 8, block (i, j) present for j in {i, i-1, i-3}: small enough for ONE row-block."""
    nb = 12
    rng = np.random.RandomState(17)
    br = [i for i in range(nb) for d in (0, 1, 3) if i - d >= 0]
    bc = [i - d for i in range(nb) for d in (0, 1, 3) if i - d >= 0]
    a8, b8 = np.meshgrid(np.arange(8), np.arange(8), indexing="ij")
    r = (np.array(br)[:, None] * 8 + a8.ravel()[None, :]).ravel()
    c = (np.array(bc)[:, None] * 8 + b8.ravel()[None, :]).ravel()
    low = sp.coo_matrix((rng.uniform(-1, 1, r.size), (r, c)), shape=(nb * 8, nb * 8)).tocsr()
    low = sp.tril(low, k=-1)
    m = (low + low.T + sp.diags(np.asarray(abs(low + low.T).sum(axis=1)).ravel() + 1.0)).tocsr()
    m.sort_indices()
    csr = (m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.copy(), nb * 8)
    opts = {"spx.preproc.sampling": "none", "spx.gpu.rowblock_elems": "8192", "spx.gpu.rowblock_rows": "512"}
    return csr, opts


def test_symmetric_tiles_inside_a_single_rowblock_need_no_spill(tmp_path):
    csr, opts = _one_rowblock_sym()
    rp, ci, va, n = csr
    A = tune(csr, opts, sym=True, host_only=True)
    f = str(tmp_path / "m.spx")
    A.save(f)
    s = Stream(f)
    assert len(s.rbs) == 1 and (s.passes["kind"][:int(s.rbs[0]["n_pass"])] == 3).any()
    assert s.n_spill == 0 and int(s.rbs[0]["n_slots"]) == 0       # every tile column is an own row
    a = sp.csr_matrix((va, ci, rp), shape=(n, n))
    x = synth.random_x(n)
    assert np.allclose(s.matvec(x), a @ x, rtol=1e-12, atol=1e-14)


@pytest.mark.gpu
def test_symmetric_tiles_inside_a_single_rowblock_gpu():
    from helpers import check_y
    csr, opts = _one_rowblock_sym()
    n = csr[3]
    A = tune(csr, opts, sym=True)
    x = synth.random_x(n)
    y = np.full(n, np.nan)
    A.matvec_mult(0.5, x, y)
    check_y(csr, x, y, 0.5)
    y0 = synth.random_x(n, seed=3)
    y = y0.copy()
    A.matvec_kernel(-1.5, x, 0.5, y)
    check_y(csr, x, y, -1.5, 0.5, y0)


@pytest.mark.parametrize("phases", ["2", "3", "5", "c2", "c4"])
def test_column_phases(tmp_path, phases):
    """spx.gpu.col_phases: the stream holds the matrix as a sum of column slices, each a run of
    row-blocks of its own (first one flagged SPX_RB_PHASE_START = 4); slice 0 covers every row,
    the others only rows that hold something of theirs.  Same product, entries found where they
    are, restored as saved."""
    csr = synth.syn_webbase(0.03)
    rp, ci, va, n = csr
    A = tune(csr, {"spx.gpu.col_phases": phases, "spx.rt.nr_threads": "2"}, host_only=True)
    f = str(tmp_path / "m.spx")
    A.save(f)
    s = Stream(f)
    starts = np.flatnonzero((s.rbs["flags"] & 4) != 0)
    concurrent = phases.startswith("c")       # one launch, a group of XCDs per slice: every row-block adds (flag 8)
    K = int(phases.lstrip("c"))
    assert len(starts) == K - 1 and starts[0] > 0
    assert bool((s.rbs["flags"] & 8).all()) == concurrent and bool((s.rbs["flags"] & 8).any()) == concurrent
    bounds = [0] + [int(v) for v in starts] + [len(s.rbs)]
    for k in range(K):
        part = s.rbs[bounds[k]:bounds[k + 1]]
        r0 = part["row0"].astype(np.int64)
        assert np.all(np.diff(r0) > 0)                                   # ascending inside a slice
        if k == 0 and not concurrent:
            assert int(part["n_rows"].astype(np.int64).sum()) == n       # slice 0 stores every row
    r, c, v, b, m = dense_of(s)
    lo = np.array([n * k // K for k in range(K + 1)])
    for k in range(K):                                                   # a slice holds its columns only
        idx = np.flatnonzero((b >= bounds[k]) & (b < bounds[k + 1]))
        assert np.all((c[idx] >= lo[k]) & (c[idx] < lo[k + 1]))
    a = sp.csr_matrix((va, ci, rp), shape=(n, n))
    assert abs(m.tocsr() - a).max() == 0
    x = synth.random_x(n)
    assert np.allclose(s.matvec(x), a @ x, rtol=1e-12, atol=1e-14)
    row = n // 2
    while rp[row] == rp[row + 1]:
        row += 1
    for k in range(rp[row], rp[row + 1]):
        assert A.get_entry(row, int(ci[k])) == va[k]
    A.set_entry(row, int(ci[rp[row]]), 4.25)
    A.save(f)
    B = sx.mat_restore(f)
    assert B.get_entry(row, int(ci[rp[row]])) == 4.25


def test_column_phases_give_way_to_overlong_rows(tmp_path):
    """A row that is split over several row-blocks is summed by a fix-up kernel that stores:
    such a matrix is emitted without phases."""
    n = 40000
    rng = np.random.RandomState(3)
    cols = np.sort(rng.choice(n, 30000, replace=False))        # > 8192 of them in every third of the columns
    rows = np.concatenate([np.zeros(cols.size, dtype=np.int64), np.arange(1, n)])
    cc = np.concatenate([cols, rng.randint(0, n, n - 1)])
    m = sp.coo_matrix((rng.uniform(-1, 1, rows.size), (rows, cc)), shape=(n, n)).tocsr()
    m.sum_duplicates(); m.sort_indices()
    csr = (m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.copy(), n)
    A = tune(csr, {"spx.gpu.col_phases": "3"}, host_only=True)
    assert A.info().n_shared_rows == 1
    f = str(tmp_path / "m.spx")
    A.save(f)
    s = Stream(f)
    assert not ((s.rbs["flags"] & 4) != 0).any()
    x = synth.random_x(n)
    assert np.allclose(s.matvec(x), m @ x, rtol=1e-12, atol=1e-14)
