"""spx_mat_save / spx_mat_restore (reference: src/api/matvec.c:409-453,
CsxSaveRestore.hpp; its round-trip test is test/src/BinaryTest_p1/p2.cpp):
the tuned matrix survives a round trip through a file without re-tuning."""
import filecmp

import numpy as np
import pytest

import sparsex_amd as sx
from sparsex_amd import synth
from helpers import tune, check_y


def test_round_trip_host_only(tmp_path):
    csr = synth.syn_cant(0.03)
    A = tune(csr, {"spx.preproc.sampling": "none", "spx.rt.nr_threads": "2"}, host_only=True)
    f1, f2 = str(tmp_path / "a.csx"), str(tmp_path / "b.csx")
    A.save(f1)
    sx.option_set("spx.rt.host_only", "true")
    B = sx.mat_restore(f1)
    ia, ib = A.info(), B.info()
    for k in ("nnz", "nnz_stored", "n_unit_elems", "n_delta_elems", "n_units", "n_rowblocks",
              "index_bytes", "value_bytes", "nr_partitions", "row_lo", "row_hi", "symmetric"):
        assert getattr(ia, k) == getattr(ib, k), k
    assert (B.nrows, B.ncols, B.nnz) == (A.nrows, A.ncols, A.nnz)
    assert A.partition() == B.partition()
    B.save(f2)
    assert filecmp.cmp(f1, f2, shallow=False)


def test_restore_rejects_garbage(tmp_path):
    p = tmp_path / "junk"
    p.write_bytes(b"not a matrix file at all")
    sx.lib().spx_log_disable_all()
    with pytest.raises(sx.SpxError):
        sx.mat_restore(str(p))
    with pytest.raises(sx.SpxError):
        sx.mat_restore(str(tmp_path / "missing"))


@pytest.mark.gpu
@pytest.mark.parametrize("sym", [False, True, "segments"])
def test_round_trip_gpu(tmp_path, sym):
    csr = synth.syn_cant(0.05)
    n = csr[3]
    o = {"spx.preproc.sampling": "none"}
    if sym == "segments":                       # read-once row segments in row-blocks of up to 2048 rows
        o.update({"spx.gpu.sym_segments": "true", "spx.gpu.sym_wide_rows": "2048"})
    A = tune(csr, o, sym=bool(sym))
    x = synth.random_x(n)
    y1 = np.zeros(n)
    A.matvec_mult(0.5, x, y1)
    f = str(tmp_path / "m.csx")
    A.save(f)
    A.destroy()
    sx.options_reset()
    B = sx.mat_restore(f)
    assert (B.info().sym_segments > 0) == (sym == "segments")
    y2 = np.full(n, np.nan)
    B.matvec_mult(0.5, x, y2)
    check_y(csr, x, y2, 0.5)
    assert np.allclose(y1, y2, rtol=1e-13, atol=1e-15)
    y0 = synth.random_x(n, seed=4)
    y3 = y0.copy()
    B.matvec_kernel(1.0, x, 2.0, y3)
    check_y(csr, x, y3, 1.0, 2.0, y0)


def test_restore_rejects_damaged_files(tmp_path):
    """A file that still carries the magic but whose index arrays were cut or
    changed must not reach the kernels (they trust every offset)."""
    csr = synth.syn_webbase(0.004)
    A = tune(csr, {"spx.rt.keep_encoded": "false"}, host_only=True)
    f = tmp_path / "m.csx"
    A.save(str(f))
    good = f.read_bytes()
    sx.option_set("spx.rt.host_only", "true")
    sx.mat_restore(str(f)).destroy()
    sx.lib().spx_log_disable_all()
    bad = tmp_path / "bad.csx"
    bad.write_bytes(good[:len(good) * 2 // 3])                      # truncated
    with pytest.raises(sx.SpxError):
        sx.mat_restore(str(bad))
    rng = np.random.RandomState(2)
    hdr = 8 + 120
    for _ in range(20):                                             # one byte of the index flipped
        b = bytearray(good)
        # (stay inside the row-block / pass / descriptor arrays that follow the header)
        pos = hdr + 8 + int(rng.randint(0, 4000))
        b[pos] ^= 0x5A
        bad.write_bytes(bytes(b))
        with pytest.raises(sx.SpxError):
            sx.mat_restore(str(bad))
