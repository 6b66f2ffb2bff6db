"""Standard Matrix Market files at size through spx_input_load_mmf (the reference's
whole input side is Matrix Market, include/sparsex/internals/Mmf.hpp:331-478 -- symmetric files
are mirrored and sorted, :445-478 -- and until now only files of a few hundred rows had been read).
The synthetic stand-ins of cant and nd24k are written the way SuiteSparse ships its matrices
(tools/mm_write.py: banner, one-based, column-ordered, lower triangle of a symmetric matrix),
loaded by the library's reader, tuned on both paths, multiplied on the GPU and checked against the
CSR product of the matrix the file was written from."""
import os
import sys

import numpy as np
import pytest

import sparsex_amd as sx
from sparsex_amd import synth
from helpers import check_y

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu

CASES = [("syn-cant", lambda: synth.syn_cant(1.0), True),          # 62 451 rows, 3.8 M nonzeros, 1.9 M lines
         ("syn-nd24k-0.25", lambda: synth.syn_nd24k(0.25), True),  # 18 000 rows, 6.9 M nonzeros
         ("syn-webbase", lambda: synth.syn_webbase(1.0), False)]   # 1 M rows, general banner


@pytest.fixture(scope="module", params=CASES, ids=[c[0] for c in CASES])
def written(request, tmp_path_factory):
    from mm_write import write_mtx
    name, gen, symmetric = request.param
    csr = gen()
    path = str(tmp_path_factory.mktemp("mtx") / (name + ".mtx"))
    entries = write_mtx(path, csr, symmetric)
    return name, csr, path, symmetric, entries


@pytest.mark.parametrize("path_sym", [False, True], ids=["general-path", "symmetric-path"])
def test_standard_file_loads_tunes_and_multiplies(written, path_sym):
    name, csr, path, symmetric, entries = written
    if path_sym and not symmetric:
        pytest.skip("unsymmetric matrix")
    rp, ci, va, n = csr
    # the file holds one triangle of a symmetric matrix; the reader mirrors it (Mmf.hpp:445-478)
    with open(path) as f:
        banner = f.readline().split()
        f.readline()
        assert f.readline().split() == [str(n), str(n), str(entries)]
    assert banner[-1] == ("symmetric" if symmetric else "general")
    assert entries == ((int(rp[-1]) + n) // 2 if symmetric else int(rp[-1]))
    sx.options_reset()
    sx.option_set("spx.rt.nr_threads", "8")
    sx.option_set("spx.matrix.symmetric", "true" if path_sym else "false")
    inp = sx.input_load_mmf(path)
    A = sx.mat_tune(inp)
    assert (A.nrows, A.ncols, A.nnz) == (n, n, int(rp[-1]))
    info = A.info()
    assert bool(info.symmetric) == path_sym and info.on_device
    x = synth.random_x(n)
    y = np.full(n, np.nan)
    A.matvec_mult(0.5, x, y)
    check_y(csr, x, y, 0.5)
    y0 = synth.random_x(n, seed=3)
    y = y0.copy()
    A.matvec_kernel(-1.5, x, 0.25, y)
    check_y(csr, x, y, -1.5, 0.25, y0)
    A.destroy()
    sx.options_reset()
