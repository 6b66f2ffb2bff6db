"""The unit windows of x (sparsex_amd/csrc/xwindows.hpp) on the CPU: what csx_spmv_xw_kernel is handed
next to the stream -- window table, translated descriptors, flagged pass headers -- is decoded here the
way the kernel decodes it (lane -> descriptor -> LDS offset -> the column the window holds there) and
must name exactly the columns the stream itself names, for every lane of every unit pass."""
import numpy as np
import pytest

import sparsex_amd as sx
from sparsex_amd import synth
from helpers import tune
from stream_decode import Stream, PASS, KIND_HORIZ, KIND_DIAG, KIND_ADIAG
from test_stream_layout import zoo

XLDS, INLINE = 2, 1

CASES = [
    ("nlpkkt", lambda: synth.syn_nlpkkt(12), {"spx.preproc.sampling": "none", "spx.rt.nr_threads": "2"}, 4096, 16),
    ("nlpkkt-small-budget", lambda: synth.syn_nlpkkt(12), {"spx.preproc.sampling": "none"}, 700, 16),
    ("nlpkkt-gap0", lambda: synth.syn_nlpkkt(10), {"spx.preproc.sampling": "none", "spx.gpu.rowblock_rows": "64"}, 4096, 0),
    ("zoo-all", zoo, {"spx.preproc.xform": "all", "spx.preproc.sampling": "none"}, 4096, 16),
    ("zoo-ad", zoo, {"spx.preproc.xform": "ad", "spx.preproc.sampling": "none", "spx.gpu.rowblock_rows": "50"}, 2048, 4),
    ("zoo-strided", zoo, {"spx.preproc.xform": "v{2},ad{3},d{2},h{4}", "spx.gpu.rowblock_rows": "23"}, 4096, 16),
    ("cant", lambda: synth.syn_cant(0.04), {"spx.preproc.sampling": "none"}, 4096, 16),
    ("cant-odd-cols", lambda: synth.syn_cant(0.031), {"spx.preproc.sampling": "none"}, 8192, 64),
    ("nd24k", lambda: synth.syn_nd24k(0.02), {"spx.preproc.sampling": "none"}, 4096, 16),
    ("webbase", lambda: synth.syn_webbase(0.02), {}, 4096, 16),
    ("band-window", lambda: synth.syn_bandrandom(8000), {}, 4096, 16),
]


def lane_columns(rb, ps, descs):
    """(first column, step of the columns per segment) of every lane of a unit pass, from `descs`."""
    nseg, mask = int(ps["nseg"]), int(ps["mask"])
    if int(ps["flags"]) & INLINE:
        mask = 0
    starts = np.array([(mask >> l) & 1 for l in range(nseg)])
    rank = int(ps["rank0"]) + np.cumsum(starts)
    d = descs[int(rb["desc_off"]) + rank]
    bits = d[:, 1].astype(np.int64)
    s = (int(ps["seg0"]) + np.arange(nseg) - ((bits >> 9) & 8191)) & 0xffff
    kind, step = (bits >> 22) & 7, bits >> 25
    dcol = np.where((kind == KIND_HORIZ) | (kind == KIND_DIAG), step, np.where(kind == KIND_ADIAG, -step, 0))
    return d[:, 0].astype(np.int64) + s * dcol, rank


@pytest.mark.parametrize("name,gen,opts,budget,gap", CASES, ids=[c[0] for c in CASES])
def test_windows_name_the_columns_of_the_stream(tmp_path, name, gen, opts, budget, gap):
    csr = gen()
    ncols = csr[3]
    A = tune(csr, opts, host_only=True)
    f = str(tmp_path / "m.spx")
    A.save(f)
    s = Stream(f)
    plan = A.unit_windows(budget, gap)
    tab, xdescs = plan["tab"], plan["xdescs"]
    xpasses = plan["passes"].view(PASS).reshape(-1)
    descs = np.stack([s.descs["col0"], s.descs["bits"]], axis=1)
    assert plan["n_rowblocks"] == len(s.rbs) and plan["n_descs"] == len(s.descs) and plan["n_passes"] == len(s.passes)
    n_win = n_unit_rb = 0
    staged = elems = elems_lds = 0
    for bi, rb in enumerate(s.rbs):
        lo = int(rb["pass_off"])
        ps_plain = s.passes[lo:lo + int(rb["n_pass"])]
        ps_x = xpasses[lo:lo + int(rb["n_pass"])]
        wins = [(int(b), int(ol) & 0xffff, int(ol) >> 16) for b, ol in tab[bi, 2:]]
        n_w = 0
        while n_w < len(wins) and wins[n_w][2]:
            n_w += 1
        assert all(w[2] == 0 for w in wins[n_w:])
        wins = wins[:n_w]
        unit = [t for t, p in enumerate(ps_plain) if p["kind"] == 0]
        n_unit_rb += bool(unit)
        elems += sum(int(ps_plain[t]["nseg"]) * int(ps_plain[t]["width"]) for t in unit)
        if not wins:
            # left alone: headers and descriptors as the stream has them
            assert (ps_x == ps_plain).all() and (tab[bi] == 0).all()
            for t in unit:
                r = int(rb["desc_off"]) + int(ps_plain[t]["rank0"])
                assert (xdescs[r] == descs[r]).all()
            continue
        n_win += 1
        # the windows: ascending, disjoint, inside the vector, even offsets, within the budget and the LDS
        total = 0
        col_at = {}
        for k, (base, off, ln) in enumerate(wins):
            assert base % 2 == 0 and off % 2 == 0 and off == total and base + ln <= ncols
            assert ln % 2 == 0 or (k == len(wins) - 1 and base + ln == ncols)
            if k:
                assert base >= wins[k - 1][0] + wins[k - 1][2]
            for i in range(ln):
                col_at[off + i] = base + i
            total += (ln + 1) & ~1
        assert total <= budget
        front = (int(rb["n_rows"]) + int(rb["xwin_len"]) + 1) & ~1
        assert front + total <= plan["lds_doubles"]
        staged += total
        # every lane of every unit pass reads, through its translated descriptor, the columns the stream names
        for t in unit:
            p, q = ps_plain[t], ps_x[t]
            assert int(q["flags"]) == int(p["flags"]) | XLDS
            for fld in ("val_off", "rank0", "seg0", "nseg", "width", "kind", "elem0"):
                assert q[fld] == p[fld]
            col, rank = lane_columns(rb, p, descs)
            xoff, xrank = lane_columns(rb, q, xdescs)
            assert (rank == xrank).all()
            W = int(p["width"])
            for w in range(W):
                got = np.array([col_at.get(int(o) + w, -1) for o in xoff])
                assert (got == col + w).all(), (name, bi, t, w)
            if int(p["flags"]) & INLINE:
                r = int(rb["desc_off"]) + int(p["rank0"])
                assert int(q["mask"]) == int(xdescs[r, 0]) | (int(xdescs[r, 1]) << 32)
            else:
                assert q["mask"] == p["mask"]
            elems_lds += int(p["nseg"]) * W
        # other passes are untouched
        for t in range(int(rb["n_pass"])):
            if t not in unit:
                assert ps_x[t] == ps_plain[t]
        # the pass range of the pipeline: unit passes of width <= 4 that read LDS, nothing else -- the longest
        # such run -- and the length of the windows (the kernel puts the pass headers behind them)
        r = int(tab[bi, 0, 0])
        a, b = r & 0xffff, r >> 16
        assert a <= b <= int(rb["n_pass"]) and int(tab[bi, 0, 1]) == total and not tab[bi, 1].any()
        narrow = [bool(ps_x[t]["kind"] == 0 and int(ps_x[t]["flags"]) & XLDS and int(ps_x[t]["width"]) <= 4)
                  for t in range(int(rb["n_pass"]))]
        assert all(narrow[a:b])
        runs, cur = [0], 0
        for ok in narrow:
            cur = cur + 1 if ok else 0
            runs.append(cur)
        assert b - a == max(runs)
        assert front + total + 3 * (s.pass_stride + 32) <= plan["lds_doubles"]
    assert n_win == plan["rowblocks_with_windows"] and n_unit_rb == plan["rowblocks_with_units"]
    assert staged == plan["staged_doubles"] and elems == plan["unit_elems"] and elems_lds == plan["unit_elems_lds"]
    if name.startswith("nlpkkt") and budget >= 4096:
        assert n_win == n_unit_rb and elems_lds == elems        # a stencil: everything fits
    if name == "nlpkkt-small-budget":
        assert 0 < n_win < n_unit_rb                             # some row-blocks do not fit 700 doubles
    sx.options_reset()


def test_no_budget_and_symmetric_streams_are_left_alone(tmp_path):
    csr = synth.syn_nlpkkt(8)
    A = tune(csr, {"spx.preproc.sampling": "none"}, host_only=True)
    plan = A.unit_windows(0, 16)
    assert plan["rowblocks_with_windows"] == 0 and not plan["tab"].any() and plan["staged_doubles"] == 0
    A = tune(csr, {"spx.preproc.sampling": "none", "spx.matrix.symmetric": "true"}, host_only=True)
    plan = A.unit_windows(4096, 16)
    assert plan["rowblocks_with_windows"] == 0 and not plan["tab"].any()
    sx.options_reset()
