"""Matrix Market reader: the SparseX header extensions and failure modes
(reference include/sparsex/internals/Mmf.hpp:331-478, :259-263)."""
import os

import numpy as np
import pytest

import sparsex_amd as sx
from sparsex_amd import synth
from helpers import oracle_y
from oracle import pyoracle


def _write(tmp_path, name, text):
    p = tmp_path / name
    p.write_text(text)
    return str(p)


BODY = [(1, 1, 1.0), (1, 3, 2.0), (2, 2, 3.0), (3, 1, 4.0), (3, 3, 5.0), (4, 2, 6.0)]


def _tune_file(path, sym=False):
    sx.options_reset()
    sx.option_set("spx.rt.host_only", "true")
    if sym:
        sx.option_set("spx.matrix.symmetric", "true")
    return sx.mat_tune(sx.input_load_mmf(path))


def _dense(path_matrix, n):
    x = np.eye(n)
    cols = []
    for j in range(n):
        y, _ = oracle_y(path_matrix, x[:, j].copy())
        cols.append(y)
    return np.array(cols).T


def test_headerless_sorted_file(tmp_path):
    text = "4 4 6\n" + "".join("%d %d %g\n" % t for t in BODY)
    A = _tune_file(_write(tmp_path, "a.mtx", text))
    D = _dense(A, 4)
    ref = np.zeros((4, 4))
    for r, c, v in BODY:
        ref[r - 1, c - 1] = v
    assert np.array_equal(D, ref)


def test_banner_general_column_major_and_zero_based(tmp_path):
    ents = sorted(BODY, key=lambda t: (t[1], t[0]))          # column-major order
    text = ("%%MatrixMarket matrix coordinate real general\n% comment\n4 4 6\n" +
            "".join("%d %d %g\n" % t for t in ents))
    D = _dense(_tune_file(_write(tmp_path, "b.mtx", text)), 4)
    text0 = ("%%MatrixMarket matrix coordinate real general 0-base row\n4 4 6\n" +
             "".join("%d %d %g\n" % (r - 1, c - 1, v) for r, c, v in BODY))
    D0 = _dense(_tune_file(_write(tmp_path, "c.mtx", text0)), 4)
    assert np.array_equal(D, D0)
    assert D[2, 0] == 4.0 and D[0, 2] == 2.0


def test_symmetric_banner_is_mirrored(tmp_path):
    text = ("%%MatrixMarket matrix coordinate real symmetric\n3 3 4\n"
            "1 1 2\n2 1 -1\n2 2 2\n3 3 5\n")
    A = _tune_file(_write(tmp_path, "s.mtx", text))
    assert A.nnz == 5                       # mirrored off-diagonal
    D = _dense(A, 3)
    assert np.array_equal(D, np.array([[2, -1, 0], [-1, 2, 0], [0, 0, 5.0]]))
    As = _tune_file(_write(tmp_path, "s.mtx", text), sym=True)
    assert np.array_equal(_dense(As, 3), D)


def test_unsorted_headerless_file_fails_cleanly(tmp_path):
    text = "4 4 6\n" + "".join("%d %d %g\n" % t for t in reversed(BODY))
    path = _write(tmp_path, "u.mtx", text)
    sx.options_reset()
    sx.option_set("spx.rt.host_only", "true")
    sx.lib().spx_log_disable_all()
    inp = sx.input_load_mmf(path)
    with pytest.raises(sx.SpxError):
        sx.mat_tune(inp)


def test_symmetric_option_on_unsymmetric_matrix_fails_cleanly(tmp_path):
    """test-sparsex.sh.in:207-214: must fail, but not by a signal."""
    import json
    from helpers import GOLDEN
    with open(os.path.join(GOLDEN, "reference_matrices.json")) as f:
        m = json.load(f)["demopatt"]          # the matrix the reference's test uses
    lines = ["%d %d %d" % (m["n"], m["n"], len(m["values"]))]
    for r in range(m["n"]):
        for j in range(m["rowptr"][r], m["rowptr"][r + 1]):
            lines.append("%d %d %r" % (r + 1, m["colind"][j] + 1, m["values"][j]))
    path = _write(tmp_path, "n.mtx", "\n".join(lines) + "\n")
    sx.options_reset()
    sx.option_set("spx.rt.host_only", "true")
    sx.option_set("spx.matrix.symmetric", "true")
    sx.lib().spx_log_disable_all()
    with pytest.raises(sx.SpxError):
        sx.mat_tune(sx.input_load_mmf(path))


def test_bad_banner_and_missing_file(tmp_path):
    sx.lib().spx_log_disable_all()
    with pytest.raises(sx.SpxError):
        sx.input_load_mmf(_write(tmp_path, "x.mtx", "%%NotMatrixMarket foo\n1 1 1\n1 1 1\n"))
    with pytest.raises(sx.SpxError):
        sx.input_load_mmf(str(tmp_path / "missing.mtx"))


def test_standard_matrix_market_file_column_major_symmetric(tmp_path):
    """What SuiteSparse ships: banner, comments, lower triangle in column-major order,
    CRLF line ends tolerated; loaded whole, mirrored and sorted (Mmf.hpp:445-478)."""
    import scipy.sparse as sp
    from sparsex_amd import synth
    from helpers import oracle_y, check_y
    rp, ci, va, n = synth.syn_cant(0.02)
    rows = np.repeat(np.arange(n), np.diff(rp))
    low = ci <= rows
    r, c, v = rows[low], ci[low], va[low]
    o = np.lexsort((r, c))
    f = tmp_path / "std.mtx"
    with open(f, "w", newline="") as fh:
        fh.write("%%MatrixMarket matrix coordinate real symmetric\r\n% a comment\r\n%another\r\n")
        fh.write("%d %d %d\r\n" % (n, n, r.size))
        for k in o:
            fh.write("%d %d %.17g\r\n" % (r[k] + 1, c[k] + 1, v[k]))
    x = synth.random_x(n)
    for sym in (False, True):
        sx.options_reset()
        sx.option_set("spx.rt.host_only", "true")
        sx.option_set("spx.rt.nr_threads", "2")
        if sym:
            sx.option_set("spx.matrix.symmetric", "true")
        A = sx.mat_tune(sx.input_load_mmf(str(f)))
        assert (A.nrows, A.ncols, A.nnz) == (n, n, rp[-1])
        yo, _ = oracle_y(A, x, 0.5)
        check_y((rp, ci, va, n), x, yo, 0.5)
