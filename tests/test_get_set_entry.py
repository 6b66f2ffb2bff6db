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
    (lambda: synth.syn_bandrandom(5000), {}, False),       # leftovers addressed through the x window
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
@pytest.mark.parametrize("sym", [False, True, "segments"])
def test_set_entry_reaches_the_gpu(sym):
    """(sym == "segments": the lower triangle in read-once row segments -- one stored value
    serves both triangles, so one poke must change both)"""
    csr = synth.syn_cant(0.03)
    rp, ci, va, n = csr
    o = {"spx.preproc.sampling": "none"}
    if sym == "segments":
        o.update({"spx.gpu.sym_segments": "true", "spx.gpu.sym_wide_rows": "2048"})
    A = tune(csr, o, sym=bool(sym))
    assert (A.info().sym_segments > 0) == (sym == "segments")
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


@pytest.mark.parametrize("sym", [False, True])
def test_get_and_set_on_a_restored_matrix_host(tmp_path, sym):
    """A restored matrix is fully usable (reference: CsxSaveRestore.hpp:316-337 rebuilds
    a complete spm_mt): entries are found in the descriptor stream itself, and the
    encoded partitions travel in the file so that the CSX export follows."""
    csr = synth.syn_nd24k(0.012) if sym else synth.syn_webbase(0.004)
    rp, ci, va, n = csr
    A = tune(csr, {"spx.preproc.sampling": "none", "spx.rt.nr_threads": "2"}, sym=sym, host_only=True)
    rows = np.repeat(np.arange(n), np.diff(rp))
    rng = np.random.RandomState(3)
    pick = rng.choice(rp[-1], size=300, replace=False)
    va2 = va.copy()
    for j in pick[:30]:                       # edits made before the save are in the file
        r, c = int(rows[j]), int(ci[j])
        A.set_entry(r, c, 1.25)
        va2[j] = 1.25
        if sym:
            va2[rp[c] + int(np.searchsorted(ci[rp[c]:rp[c + 1]], r))] = 1.25
    f = str(tmp_path / "m.csx")
    A.save(f)
    A.destroy()
    sx.options_reset()
    sx.option_set("spx.rt.host_only", "true")
    B = sx.mat_restore(f)
    for j in pick:
        assert B.get_entry(int(rows[j]), int(ci[j])) == va2[j]
    for j in pick[30:60]:
        r, c = int(rows[j]), int(ci[j])
        B.set_entry(r, c, -0.75)
        va2[j] = -0.75
        if sym:
            va2[rp[c] + int(np.searchsorted(ci[rp[c]:rp[c + 1]], r))] = -0.75
    x = synth.random_x(n)
    yo, _ = oracle_y(B, x, 1.0)               # CSX export of the restored matrix
    check_y((rp, ci, va2, n), x, yo, 1.0)


def test_entries_without_the_encoded_partitions_host():
    csr = synth.syn_cant(0.02)
    rp, ci, va, n = csr
    A = tune(csr, {"spx.preproc.sampling": "none", "spx.rt.keep_encoded": "false"}, host_only=True)
    rows = np.repeat(np.arange(n), np.diff(rp))
    for j in np.random.RandomState(1).choice(rp[-1], size=500, replace=False):
        assert A.get_entry(int(rows[j]), int(ci[j])) == va[j]
    A.set_entry(int(rows[7]), int(ci[7]), 9.0)
    assert A.get_entry(int(rows[7]), int(ci[7])) == 9.0


@pytest.mark.gpu
@pytest.mark.parametrize("sym", [False, True])
def test_set_save_destroy_restore_multiply_gpu(tmp_path, sym):
    """set_entry -> save -> destroy -> restore -> SpMV: the file holds the edit, the
    restored handle multiplies with it, and entries can be read and changed again."""
    csr = synth.syn_nd24k(0.02) if sym else synth.syn_cant(0.03)
    rp, ci, va, n = csr
    A = tune(csr, {}, sym=sym)
    rows = np.repeat(np.arange(n), np.diff(rp))
    rng = np.random.RandomState(8)
    va2 = va.copy()

    def edit(M, js, val):
        for j in js:
            r, c = int(rows[j]), int(ci[j])
            M.set_entry(r, c, val)
            va2[j] = val
            if sym:
                va2[rp[c] + int(np.searchsorted(ci[rp[c]:rp[c + 1]], r))] = val
    pick = rng.choice(rp[-1], size=60, replace=False)
    edit(A, pick[:20], 2.5)
    f = str(tmp_path / "m.csx")
    A.save(f)
    A.destroy()
    sx.options_reset()
    B = sx.mat_restore(f)
    x = synth.random_x(n)
    y = np.full(n, np.nan)
    B.matvec_mult(1.0, x, y)
    check_y((rp, ci, va2, n), x, y, 1.0)
    for j in pick:
        assert B.get_entry(int(rows[j]), int(ci[j])) == va2[j]
    edit(B, pick[20:40], -1.5)
    B.matvec_mult(1.0, x, y)
    check_y((rp, ci, va2, n), x, y, 1.0)
