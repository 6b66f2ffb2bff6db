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


# ---- the parallel reader (round 6): pieces cut at line ends, all host threads, CSR arrays -----------------------------

def _csr_of(A):
    """the matrix a tuned handle multiplies with, as a dense-free check: y = A x for a few x against scipy"""
    return A


def _write_big(tmp_path, name, banner, n, r, c, v, order=None, fmt="%d %d %.17g", tail=""):
    p = tmp_path / name
    if order is not None:
        r, c, v = r[order], c[order], v[order]
    with open(p, "w") as f:
        f.write(banner)
        f.write("%d %d %d\n" % (n, n, r.size))
        np.savetxt(f, np.column_stack([r, c, v]), fmt=fmt)
        f.write(tail)
    return str(p)


@pytest.fixture(scope="module")
def big():
    rp, ci, va, n = synth.syn_nlpkkt(26)             # 35 152 rows, 0.9 M nonzeros: a file of some 30 MB = several pieces
    rp, ci, va = np.asarray(rp), np.asarray(ci), np.asarray(va)
    rows = np.repeat(np.arange(n), np.diff(rp))
    return (rp, ci, va, n), rows


def _same_product(A, csr, sym=False):
    from helpers import check_y
    x = synth.random_x(csr[3], seed=9)
    y, _ = oracle_y(A, x)
    check_y(csr, x, y, 1.0)


def test_large_files_in_pieces_general_and_symmetric(tmp_path, big):
    csr, rows = big
    rp, ci, va, n = csr
    # standard general file, column-major (what SuiteSparse ships), 1-based
    o = np.lexsort((rows, ci))
    path = _write_big(tmp_path, "g.mtx", "%%MatrixMarket matrix coordinate real general\n% comment\n", n, rows + 1, ci + 1, va, o)
    assert os.path.getsize(path) > 3 * (8 << 20)
    A = _tune_file(path)
    assert A.nnz == len(ci)
    _same_product(A, csr)
    # the same entries shuffled, zero-based, CRLF-free, no newline at the end of the file
    rng = np.random.default_rng(5)
    sh = rng.permutation(len(ci))
    path = _write_big(tmp_path, "h.mtx", "%%MatrixMarket matrix coordinate real general 0-base\n", n, rows, ci, va, sh)
    with open(path, "rb+") as f:
        f.seek(-1, 2)
        assert f.read(1) == b"\n"
        f.seek(-1, 2)
        f.truncate()
    _same_product(_tune_file(path), csr)
    # symmetric file: lower triangle, column-major; general and symmetric tuning
    low = ci <= rows
    r, c, v = rows[low], ci[low], va[low]
    path = _write_big(tmp_path, "s.mtx", "%%MatrixMarket matrix coordinate real symmetric\n", n, r + 1, c + 1, v, np.lexsort((r, c)))
    A = _tune_file(path)
    assert A.nnz == len(ci)
    _same_product(A, csr)
    _same_product(_tune_file(path, sym=True), csr)
    # row-major with the `row` keyword: read when the tuner asks, must be sorted
    path = _write_big(tmp_path, "r.mtx", "%%MatrixMarket matrix coordinate real general row\n", n, rows + 1, ci + 1, va)
    _same_product(_tune_file(path), csr)


def test_large_file_failure_modes(tmp_path, big):
    csr, rows = big
    rp, ci, va, n = csr
    sx.lib().spx_log_disable_all()
    o = np.lexsort((rows, ci))
    banner = "%%MatrixMarket matrix coordinate real general\n"
    # whatever follows the entries the size line claims is not looked at
    ok = _write_big(tmp_path, "t.mtx", banner, n, rows + 1, ci + 1, va, o, tail="this is not an entry\n\n\n")
    _same_product(_tune_file(ok), csr)
    # ... a line among them that does not parse is an error, wherever in the file it lies (here: in the last piece)
    text = open(ok).read().split("\n")
    text[-20] = "17 oops 1.0"
    bad = tmp_path / "bad.mtx"
    bad.write_text("\n".join(text))
    with pytest.raises(sx.SpxError):
        sx.input_load_mmf(str(bad))
    # fewer entries than claimed
    short = tmp_path / "short.mtx"
    short.write_text("\n".join(text[:len(text) // 2]) + "\n")
    with pytest.raises(sx.SpxError):
        sx.input_load_mmf(str(short))
    # an entry outside the matrix
    text = open(ok).read().split("\n")
    text[len(text) // 2] = "%d 1 1.0" % (n + 1)
    out = tmp_path / "out.mtx"
    out.write_text("\n".join(text))
    with pytest.raises(sx.SpxError):
        sx.input_load_mmf(str(out))
    # ... also one whose coordinate does not fit the index type (it must not wrap into the matrix)
    text[len(text) // 2] = "%d 1 1.0" % (2 ** 32 + 1)
    out.write_text("\n".join(text))
    with pytest.raises(sx.SpxError):
        sx.input_load_mmf(str(out))
    # a row-major file that is not sorted fails when the tuner reads it (as the reference: Mmf.hpp:259-263)
    unsorted = _write_big(tmp_path, "u.mtx", "%%MatrixMarket matrix coordinate real general row\n", n, rows + 1, ci + 1, va, o)
    inp = sx.input_load_mmf(unsorted)
    sx.options_reset()
    sx.option_set("spx.rt.host_only", "true")
    with pytest.raises(sx.SpxError):
        sx.mat_tune(inp)
    sx.lib().spx_log_error_console()


def test_number_forms_duplicates_and_empty(tmp_path):
    # what strtol / strtod accept: signs, exponents, a leading '+', leading blanks, tabs, hexadecimal
    text = ("%%MatrixMarket matrix coordinate real general\n3 3 6\n"
            "  1 1 +2.5\n1\t3\t-1e-1\n2 2 .5\n+3 1 4E0\n3 3 0x1p2\n2 1 7\n")
    A = _tune_file(_write(tmp_path, "f.mtx", text))
    assert np.array_equal(_dense(A, 3), np.array([[2.5, 0, -0.1], [7, 0.5, 0], [4, 0, 4.0]]))
    # an entry given twice stays two elements (their products add up)
    text = "%%MatrixMarket matrix coordinate real general\n2 2 3\n1 1 1\n2 2 2\n1 1 0.5\n"
    A = _tune_file(_write(tmp_path, "d.mtx", text))
    assert A.nnz == 3
    assert np.array_equal(_dense(A, 2), np.array([[1.5, 0], [0, 2.0]]))
    # no entries at all
    sx.lib().spx_log_disable_all()
    inp = sx.input_load_mmf(_write(tmp_path, "e.mtx", "%%MatrixMarket matrix coordinate real general\n4 4 0\n"))
    sx.lib().spx_log_error_console()
    assert inp is not None
