"""GPU parity: the HIP path (through the C ABI) against the oracle.

Every case multiplies through libsparsex.so's spx_matvec_* on a real MI355X
and compares with (a) the oracle decoding the very same tuned matrix in the
reference's CSX byte format and (b) the reference tests' CSR criterion
(test/src/CsxCheck.cpp:28-48, relative 1e-6) plus the stated fp64 bound.
"""
import numpy as np
import pytest

import sparsex_amd as sx
from sparsex_amd import synth
from helpers import tune, oracle_y, check_y, check_vs_oracle

pytestmark = pytest.mark.gpu

CASES = [
    ("cant", lambda: synth.syn_cant(0.05)),
    ("web", lambda: synth.syn_webbase(0.02)),
    ("nd24k", lambda: synth.syn_nd24k(0.02)),
    ("kkt", lambda: synth.syn_nlpkkt(8)),
    ("kkt2f", lambda: synth.syn_kkt2f(7)),
    ("band", lambda: synth.syn_bandrandom(20000)),        # leftovers gather from the LDS x window
]
OPTS = [
    {},
    {"spx.preproc.sampling": "none"},
    {"spx.preproc.sampling": "none", "spx.preproc.xform": "h,v,d,ad"},
    {"spx.preproc.sampling": "none", "spx.rt.nr_threads": "3"},
    {"spx.preproc.sampling": "none", "spx.gpu.rowblock_elems": "300",
     "spx.gpu.rowblock_rows": "7"},
    {"spx.gpu.rowblock_elems": "8000", "spx.gpu.rowblock_rows": "2048"},     # row-blocks joined from planned ones
]


@pytest.mark.parametrize("name,gen", CASES)
@pytest.mark.parametrize("opts", OPTS)
def test_mult_general(name, gen, opts):
    csr = gen()
    A = tune(csr, opts)
    n = csr[3]
    x = synth.random_x(n)
    y = np.full(n, np.nan)          # mult must overwrite, never read, y
    A.matvec_mult(0.5, x, y)
    yo, _ = oracle_y(A, x, 0.5)
    check_vs_oracle(csr, x, y, yo, 0.5)
    check_y(csr, x, y, 0.5)
    check_y(csr, x, yo, 0.5)
    check_vs_oracle(csr, x, y, yo, 0.5)
    # repeated calls agree to rounding (the wavefronts of a row-block add into
    # its LDS tile in no fixed order) and never depend on y's previous contents
    y2 = np.zeros(n)
    A.matvec_mult(0.5, x, y2)
    check_y(csr, x, y2, 0.5)


@pytest.mark.parametrize("name,gen", CASES)
def test_kernel_beta(name, gen):
    csr = gen()
    A = tune(csr, {"spx.preproc.sampling": "none"})
    n = csr[3]
    x = synth.random_x(n)
    y0 = synth.random_x(n, seed=7)
    y = y0.copy()
    A.matvec_kernel(1.5, x, -0.25, y)
    check_y(csr, x, y, 1.5, -0.25, y0)


@pytest.mark.parametrize("name,gen", [c for c in CASES if c[0] not in ("web", "band")])
@pytest.mark.parametrize("opts", [{}, {"spx.preproc.sampling": "none"},
                                  {"spx.preproc.sampling": "none", "spx.rt.nr_threads": "4"}])
def test_mult_symmetric(name, gen, opts):
    csr = gen()
    A = tune(csr, opts, sym=True)
    n = csr[3]
    x = synth.random_x(n)
    y = np.full(n, np.nan)
    A.matvec_mult(0.5, x, y)
    yo, _ = oracle_y(A, x, 0.5)
    check_vs_oracle(csr, x, y, yo, 0.5)
    check_y(csr, x, y, 0.5)
    check_y(csr, x, yo, 0.5)
    y0 = synth.random_x(n, seed=9)
    y = y0.copy()
    A.matvec_kernel(2.0, x, 0.5, y)
    check_y(csr, x, y, 2.0, 0.5, y0)


def test_long_rows_shared():
    """Rows longer than a row-block are chunked and summed by the fix-up kernel."""
    rng = np.random.RandomState(3)
    n = 40000
    import scipy.sparse as sp
    rows = np.concatenate([np.full(30000, 5), np.full(9000, 17), rng.randint(0, n, 50000)])
    cols = np.concatenate([rng.choice(n, 30000, replace=False),
                           rng.choice(n, 9000, replace=False), rng.randint(0, n, 50000)])
    a = sp.coo_matrix((np.ones(rows.size), (rows, cols)), shape=(n, n)).tocsr()
    a.sum_duplicates(); a.sort_indices()
    a.data = rng.uniform(-1, 1, a.nnz)
    csr = (a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data, n)
    A = tune(csr, {})
    assert A.info().n_shared_rows >= 2
    x = synth.random_x(n)
    y = np.full(n, np.nan)
    A.matvec_mult(1.0, x, y)
    check_y(csr, x, y, 1.0)
    y0 = synth.random_x(n, seed=5)
    y = y0.copy()
    A.matvec_kernel(0.5, x, 2.0, y)
    check_y(csr, x, y, 0.5, 2.0, y0)


@pytest.mark.parametrize("segs", ["true", "false"])
def test_symmetric_long_rows(segs):
    """Symmetric storage with rows beyond a row-block (shared rows, fix-up kernel after the atomic
    hand-over), with and without read-once segments in the other rows."""
    from test_stream_layout import long_row_sym
    csr, m = long_row_sym()
    n = csr[3]
    A = tune(csr, {"spx.gpu.sym_segments": segs, "spx.rt.nr_threads": "3"}, sym=True)
    assert A.info().n_shared_rows >= 2
    x = synth.random_x(n)
    y = np.full(n, np.nan)
    A.matvec_mult(1.0, x, y)
    check_y(csr, x, y, 1.0)
    y0 = synth.random_x(n, seed=5)
    y = y0.copy()
    A.matvec_kernel(0.5, x, 2.0, y)
    check_y(csr, x, y, 0.5, 2.0, y0)


@pytest.mark.parametrize("opts", [{"spx.gpu.sym_spill": "lists"}, {"spx.gpu.sym_spill": "atomic"},
                                  {"spx.gpu.deterministic": "true"}], ids=["lists", "atomic", "deterministic"])
def test_symmetric_long_row_that_tiles_add_to(opts):
    """A row too long for one row-block (its value comes from the fix-up kernel, by a store) that
    is also a column of dense tiles further down (their transposed sums are added to it): the
    spilled sums must come after the fix-up (found by tools/soak_large.py: they were overwritten)."""
    import scipy.sparse as sp
    n, R = 30000, 16400
    rng = np.random.RandomState(12)
    r1 = np.full(9000, R); c1 = np.sort(rng.choice(R - 8, 9000, replace=False))
    a8, b8 = np.meshgrid(np.arange(8), np.arange(8), indexing="ij")
    tr = (8 * np.arange(2100, 2140)[:, None] + a8.ravel()[None, :]).ravel()       # tiles below the long row ...
    tc = np.tile(R + b8.ravel(), 40)                                              # ... in its column block
    rb = np.arange(1, n); cb = rb - 1
    r, c = np.concatenate([r1, tr, rb]), np.concatenate([c1, tc, cb])
    low = sp.coo_matrix((rng.uniform(0.5, 1.5, r.size), (r, c)), shape=(n, n)).tocsr()
    low.sum_duplicates()
    low = sp.tril(low, k=-1)
    m = (low + low.T + sp.diags(rng.uniform(1.0, 2.0, n))).tocsr()
    m.sort_indices()
    csr = (m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.copy(), n)
    A = tune(csr, dict(opts), sym=True)
    assert A.info().n_shared_rows >= 1 and A.info().sym_tiles in (1, 2)
    x = synth.random_x(n)
    y = np.full(n, np.nan)
    A.matvec_mult(0.5, x, y)
    check_y(csr, x, y, 0.5)
    y0 = synth.random_x(n, seed=5)
    y = y0.copy()
    A.matvec_kernel(2.0, x, -0.5, y)
    check_y(csr, x, y, 2.0, -0.5, y0)


def test_demopatt_reference_scenarios():
    """The reference's own scenario list on its fixtures, 128 loops, alpha 0.5
    (test/scripts/test-sparsex.sh.in:55-244, test/src/sparsex_test.c:57-191)."""
    import os
    from helpers import GOLDEN
    import json
    with open(os.path.join(GOLDEN, "reference_matrices.json")) as f:
        mats = json.load(f)
    scen = [
        ("demopatt", {}, False), ("demopatt", {"spx.preproc.xform": "h"}, False),
        ("demopatt", {"spx.preproc.xform": "v"}, False),
        ("demopatt", {"spx.preproc.xform": "all"}, False),
        ("symmetric", {"spx.preproc.xform": "all", "spx.preproc.sampling": "portion",
                       "spx.preproc.sampling.nr_samples": "2",
                       "spx.preproc.sampling.portion": "0.4"}, True),
        ("demopatt", {"spx.rt.nr_threads": "2", "spx.rt.cpu_affinity": "0,1",
                      "spx.preproc.xform": "all"}, False),
        ("symmetric", {"spx.preproc.xform": "all"}, True),
        ("symmetric-very-sparse", {"spx.preproc.xform": "all"}, True),
        ("symmetric", {"spx.rt.nr_threads": "2", "spx.rt.cpu_affinity": "0,1",
                       "spx.preproc.xform": "all"}, True),
    ]
    for name, opts, sym in scen:
        m = mats[name]
        csr = (np.array(m["rowptr"], dtype=np.int32), np.array(m["colind"], dtype=np.int32),
               np.array(m["values"]), m["n"])
        A = tune(csr, opts, sym=sym)
        x = synth.random_x(m["n"])
        y = np.zeros(m["n"])
        for _ in range(128):
            A.matvec_mult(0.5, x, y)
        check_y(csr, x, y, 0.5)


@pytest.mark.parametrize("gen", ["kkt2f", "nlpkkt"])
@pytest.mark.parametrize("wide", ["512", "1024", "2048"])
def test_symmetric_segments_in_wide_rowblocks(wide, gen):
    """spx.gpu.sym_wide_rows: row-blocks of up to 2048 rows (several planned ones side by side,
    common slots) give the same product."""
    csr = synth.syn_kkt2f_rows(36) if gen == "kkt2f" else synth.syn_nlpkkt_rows(40)
    n = csr[3]
    A = tune(csr, {"spx.gpu.sym_segments": "true", "spx.gpu.sym_wide_rows": wide, "spx.rt.nr_threads": "4"}, sym=True)
    assert A.info().sym_segments == 2
    x = synth.random_x(n)
    y = np.full(n, np.nan)
    A.matvec_mult(0.5, x, y)
    check_y(csr, x, y, 0.5)
    yo, _ = oracle_y(A, x, 0.5)
    check_vs_oracle(csr, x, y, yo, 0.5)
    y0 = synth.random_x(n, seed=3)
    y1 = y0.copy()
    A.matvec_kernel(1.5, x, -0.25, y1)
    check_y(csr, x, y1, 1.5, -0.25, y0)


@pytest.mark.parametrize("segs", ["true", "false"])
@pytest.mark.parametrize("threads", ["1", "3"])
@pytest.mark.parametrize("name,gen", [
    ("cant", lambda: synth.syn_cant(0.1)),
    ("nd24k", lambda: synth.syn_nd24k(0.05)),
    ("kkt2f", lambda: synth.syn_kkt2f_rows(20)),
    ("nlpkkt", lambda: synth.syn_nlpkkt_rows(24)),
], ids=["cant", "nd24k", "kkt2f", "nlpkkt"])
def test_symmetric_read_once_segments(name, gen, threads, segs):
    """spx.gpu.sym_segments: runs of consecutive columns of the lower triangle are read once --
    the lane adds its row sum to the y tile and value * x[row] to the columns' rows (LDS slots
    or global atomics); same product as with the mirror image stored."""
    csr = gen()
    n = csr[3]
    A = tune(csr, {"spx.gpu.sym_segments": segs, "spx.rt.nr_threads": threads}, sym=True)
    assert (A.info().sym_segments > 0) == (segs == "true")
    x = synth.random_x(n)
    y = np.full(n, np.nan)
    A.matvec_mult(0.5, x, y)
    check_y(csr, x, y, 0.5)
    yo, _ = oracle_y(A, x, 0.5)
    check_vs_oracle(csr, x, y, yo, 0.5)
    y0 = synth.random_x(n, seed=3)
    y1 = y0.copy()
    A.matvec_kernel(1.5, x, -0.25, y1)
    check_y(csr, x, y1, 1.5, -0.25, y0)
    # repeated products agree to rounding (the hand-over order varies)
    y2 = np.empty(n)
    A.matvec_mult(0.5, x, y2)
    check_y(csr, x, y2, 0.5)


@pytest.mark.parametrize("mode", ["lists", "atomic", "auto"])
@pytest.mark.parametrize("threads", ["1", "3"])
def test_symmetric_tiles_hand_over_modes(mode, threads):
    """The transposed sums of the symmetric tiles reach their rows through the spill array
    and a second kernel (fixed order) or straight through global atomics
    (spx.gpu.sym_spill); same product either way, with and without beta."""
    csr = synth.syn_nd24k(0.05)
    n = csr[3]
    A = tune(csr, {"spx.gpu.sym_spill": mode, "spx.rt.nr_threads": threads}, sym=True)
    assert A.info().sym_tiles == {"lists": 1, "atomic": 2}.get(mode, A.info().sym_tiles) and A.info().sym_tiles in (1, 2)
    x = synth.random_x(n)
    y = np.full(n, np.nan)
    A.matvec_mult(0.5, x, y)
    check_y(csr, x, y, 0.5)
    yo, _ = oracle_y(A, x, 0.5)
    check_vs_oracle(csr, x, y, yo, 0.5)
    y0 = synth.random_x(n, seed=3)
    y1 = y0.copy()
    A.matvec_kernel(1.5, x, -0.25, y1)
    check_y(csr, x, y1, 1.5, -0.25, y0)


@pytest.mark.parametrize("window", ["true", "false"])
def test_x_window_on_and_off(window):
    """Leftovers whose columns lie close together gather from a window of x staged in LDS
    (SPX_PASS_GATHER_LDS), the others through L2: same product with the window switched off."""
    csr = synth.syn_bandrandom(30000, band=200, per_row=30)
    n = csr[3]
    A = tune(csr, {"spx.gpu.x_window": window, "spx.rt.nr_threads": "2"})
    x = synth.random_x(n)
    y = np.full(n, np.nan)
    A.matvec_mult(0.5, x, y)
    check_y(csr, x, y, 0.5)
    yo, _ = oracle_y(A, x, 0.5)
    check_vs_oracle(csr, x, y, yo, 0.5)


@pytest.mark.parametrize("name,gen,sym", [
    ("cant", lambda: synth.syn_cant(0.3), False),
    ("web", lambda: synth.syn_webbase(0.2), False),
    ("nd24k-sym", lambda: synth.syn_nd24k(0.1), True),
    ("cant-sym", lambda: synth.syn_cant(0.2), True),
    ("band", lambda: synth.syn_bandrandom(40000), False),
], ids=["cant", "web", "nd24k-sym", "cant-sym", "band"])
def test_deterministic_mode_is_bitwise_repeatable(name, gen, sym):
    """spx.gpu.deterministic: every wavefront adds into a y tile of its own and the copies are
    summed in wavefront order; the symmetric tiles' sums travel through the fixed-order lists.
    Repeated products are then bit-identical (the reference's CPU kernels are deterministic too)."""
    csr = gen()
    n = csr[3]
    A = tune(csr, {"spx.gpu.deterministic": "true", "spx.rt.nr_threads": "2"}, sym=sym)
    x = synth.random_x(n)
    y0 = np.full(n, np.nan)
    A.matvec_mult(0.5, x, y0)
    check_y(csr, x, y0, 0.5)
    for _ in range(8):
        y = np.full(n, np.nan)
        A.matvec_mult(0.5, x, y)
        assert np.array_equal(y, y0)
    yb = synth.random_x(n, seed=9)
    y1 = yb.copy()
    A.matvec_kernel(1.5, x, -0.5, y1)
    check_y(csr, x, y1, 1.5, -0.5, yb)
    for _ in range(4):
        y2 = yb.copy()
        A.matvec_kernel(1.5, x, -0.5, y2)
        assert np.array_equal(y1, y2)


@pytest.mark.parametrize("phases", ["2", "3", "c2", "c4", "c8", "auto"])
@pytest.mark.parametrize("name,gen", [
    ("web", lambda: synth.syn_webbase(0.1)),
    ("band", lambda: synth.syn_bandrandom(30000)),
    ("cant", lambda: synth.syn_cant(0.05)),
], ids=["web", "band", "cant"])
def test_column_phases(name, gen, phases):
    """spx.gpu.col_phases: column slices launched one after the other (slice k > 0 adds to y):
    same product, alpha/beta kernel included."""
    csr = gen()
    n = csr[3]
    A = tune(csr, {"spx.gpu.col_phases": phases, "spx.rt.nr_threads": "2"})
    x = synth.random_x(n)
    y = np.full(n, np.nan)
    A.matvec_mult(0.5, x, y)
    check_y(csr, x, y, 0.5)
    yo, _ = oracle_y(A, x, 0.5)
    check_vs_oracle(csr, x, y, yo, 0.5)
    y0 = synth.random_x(n, seed=3)
    y1 = y0.copy()
    A.matvec_kernel(1.5, x, -0.25, y1)
    check_y(csr, x, y1, 1.5, -0.25, y0)


def test_band_launch_order_keeps_the_product_and_the_stream(tmp_path):
    """spx.gpu.band_order: where the rows read x in bands that recur at a fixed row distance (a 3-D
    stencil), the row-blocks of every XCD's part go up in another launch order (device copy only).
    The product is the same, entries are found and set where they are, and a saved matrix holds the
    stream in its own order again."""
    csr = synth.syn_nlpkkt_rows(48)
    rp, ci, va, n = csr
    va = va.copy()
    opts = {"spx.gpu.band_order": "true", "spx.gpu.rowblock_elems": "700", "spx.rt.nr_threads": "4",
            "spx.gpu.waves": "4", "spx.gpu.wave_tiles": "false"}      # (pinned: nothing the launch tuner measures goes into the file)
    A = tune((rp, ci, va, n), opts)
    P = tune((rp, ci, va, n), dict(opts, **{"spx.gpu.band_order": "false"}))
    assert A.info().n_rowblocks == P.info().n_rowblocks > 4096
    x = synth.random_x(n)
    y = np.full(n, np.nan)
    A.matvec_mult(0.5, x, y)
    check_y((rp, ci, va, n), x, y, 0.5)
    row = n - 1000
    k = int(rp[row])
    assert A.get_entry(row, int(ci[k])) == va[k]
    A.set_entry(row, int(ci[k]), 3.5)
    va[k] = 3.5
    fa, fp = str(tmp_path / "a.spx"), str(tmp_path / "p.spx")
    A.save(fa)
    P.set_entry(row, int(ci[k]), 3.5)
    P.save(fp)
    # same bytes as the matrix that was never reordered (the order is the device's business)
    assert open(fa, "rb").read() == open(fp, "rb").read()
    B = sx.mat_restore(fa)
    y2 = np.full(n, np.nan)
    B.matvec_mult(0.5, x, y2)
    check_y((rp, ci, va, n), x, y2, 0.5)
    assert B.get_entry(row, int(ci[k])) == 3.5


@pytest.mark.parametrize("parts", [2, 4, 7])
def test_product_in_parts(parts):
    """spx_hip_matvec_parts: the product in several launches over consecutive parts of the rows (what
    the overlapped multi-GPU step interleaves its exchange rounds with): the same y, beta included."""
    import torch
    csr = synth.syn_nlpkkt_rows(40)
    rp, ci, va, n = csr
    A = tune(csr, {"spx.gpu.rowblock_elems": "1024", "spx.rt.nr_threads": "4"})
    xh = synth.random_x(n)
    x = torch.from_numpy(xh).cuda()
    y0 = synth.random_x(n, seed=9)
    y = torch.from_numpy(y0.copy()).cuda()
    k = A.hip_matvec_parts(0.5, x.data_ptr(), -0.75, y.data_ptr(), parts, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert k == parts
    check_y(csr, xh, y.cpu().numpy(), 0.5, -0.75, y0)
    # a symmetric matrix is not cut: one launch, same product
    S = tune(csr, {"spx.rt.nr_threads": "4"}, sym=True)
    y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    assert S.hip_matvec_parts(0.5, x.data_ptr(), 0.0, y.data_ptr(), parts, torch.cuda.current_stream().cuda_stream) == 1
    torch.cuda.synchronize()
    check_y(csr, xh, y.cpu().numpy(), 0.5)
