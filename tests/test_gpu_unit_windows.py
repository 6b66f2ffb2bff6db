"""GPU parity of the general kernel with the unit windows of x in LDS and pipelined unit passes
(csx_spmv_xw_kernel, spx.gpu.unit_windows): through the C ABI against the oracle decoding the same tuned
matrix and against CSR, on every unit type, with budgets that leave some or all row-blocks on the plain
path, through the beta path, save / restore, set-entry and the cut product."""
import numpy as np
import pytest
import torch

import sparsex_amd as sx
from sparsex_amd import synth
from helpers import tune, oracle_y, check_y, check_vs_oracle
from test_stream_layout import zoo

pytestmark = pytest.mark.gpu

ON = {"spx.gpu.unit_windows": "true"}

CASES = [
    ("kkt", lambda: synth.syn_nlpkkt(16)),
    ("kkt-odd", lambda: synth.syn_nlpkkt(11)),
    ("kkt2f", lambda: synth.syn_kkt2f(9)),
    ("cant", lambda: synth.syn_cant(0.06)),
    ("nd24k", lambda: synth.syn_nd24k(0.03)),
    ("web", lambda: synth.syn_webbase(0.03)),
    ("band", lambda: synth.syn_bandrandom(20000)),
    ("zoo", zoo),
]
OPTS = [
    {},
    {"spx.preproc.sampling": "none"},
    {"spx.preproc.sampling": "none", "spx.preproc.xform": "h,v,d,ad", "spx.gpu.waves": "8"},
    {"spx.preproc.sampling": "none", "spx.rt.nr_threads": "3", "spx.gpu.waves": "2"},
    {"spx.preproc.sampling": "none", "spx.gpu.rowblock_elems": "300", "spx.gpu.rowblock_rows": "7", "spx.gpu.waves": "4"},
    {"spx.gpu.rowblock_elems": "8000", "spx.gpu.unit_window_doubles": "12000"},        # (> 64 KB of LDS)
    {"spx.preproc.sampling": "none", "spx.gpu.unit_window_doubles": "600"},            # some row-blocks do not fit
    {"spx.preproc.sampling": "none", "spx.gpu.unit_window_gap": "0", "spx.gpu.stack_segments": "false"},
    {"spx.preproc.xform": "none"},                                                     # no mined units: leftovers only
    # row-blocks joined from planned ones (their passes carry the first row of their part)
    {"spx.gpu.rowblock_elems": "20000", "spx.gpu.rowblock_rows": "2048", "spx.gpu.unit_window_doubles": "12000"},
    {"spx.preproc.sampling": "none", "spx.gpu.rowblock_elems": "1200", "spx.gpu.rowblock_rows": "1024", "spx.gpu.waves": "8"},
]


@pytest.mark.parametrize("name,gen", CASES)
@pytest.mark.parametrize("opts", OPTS)
def test_mult_with_unit_windows(name, gen, opts):
    csr = gen()
    A = tune(csr, dict(opts, **ON))
    n = csr[3]
    inf = A.info()
    x = synth.random_x(n)
    y = np.full(n, np.nan)
    A.matvec_mult(0.5, x, y)
    yo, _ = oracle_y(A, x, 0.5)
    check_vs_oracle(csr, x, y, yo, 0.5)
    check_y(csr, x, y, 0.5)
    y2 = np.zeros(n)
    A.matvec_mult(0.5, x, y2)
    check_y(csr, x, y2, 0.5)
    # the kernel under test is the one that ran wherever some row-block's columns fit
    if inf.unit_window_lds:
        assert inf.unit_windows == 1 and inf.unit_window_elems > 0
    sx.options_reset()


def test_the_stencil_runs_from_the_windows():
    csr = synth.syn_nlpkkt(20)
    A = tune(csr, dict({"spx.preproc.sampling": "none", "spx.gpu.unit_window_doubles": "4096"}, **ON))
    inf = A.info()
    assert inf.unit_windows == 1 and inf.unit_window_elems == inf.n_unit_elems > 0.9 * inf.nnz_stored
    # ... and nearly so with the default budget (a few row-blocks across the seams of the KKT blocks do not fit)
    A = tune(csr, dict({"spx.preproc.sampling": "none"}, **ON))
    inf = A.info()
    assert inf.unit_windows == 1 and inf.unit_window_elems > 0.9 * inf.n_unit_elems
    n = csr[3]
    x = synth.random_x(n)
    y = np.full(n, np.nan)
    A.matvec_mult(1.0, x, y)
    check_y(csr, x, y, 1.0)
    sx.options_reset()


@pytest.mark.parametrize("name,gen", CASES[:5])
def test_kernel_beta_with_unit_windows(name, gen):
    csr = gen()
    A = tune(csr, dict({"spx.preproc.sampling": "none"}, **ON))
    n = csr[3]
    x = synth.random_x(n)
    y0 = synth.random_x(n, seed=7)
    y = y0.copy()
    A.matvec_kernel(1.5, x, -0.25, y)
    check_y(csr, x, y, 1.5, -0.25, y0)
    sx.options_reset()


def test_auto_measures_and_is_right_either_way():
    for gen in (lambda: synth.syn_nlpkkt(30), lambda: synth.syn_cant(0.5), lambda: synth.syn_webbase(0.3)):
        csr = gen()
        A = tune(csr, {})
        n = csr[3]
        x = synth.random_x(n)
        y = np.full(n, np.nan)
        A.matvec_mult(0.5, x, y)
        check_y(csr, x, y, 0.5)
        assert A.info().unit_windows in (0, 1)
    sx.options_reset()


def test_save_restore_set_entry_and_parts(tmp_path):
    csr = synth.syn_nlpkkt(18)
    rp, ci, va, n = csr
    A = tune(csr, dict({"spx.preproc.sampling": "none"}, **ON))
    assert A.info().unit_windows == 1
    x = synth.random_x(n)
    # a value changed in HBM is seen by the kernel (the windows hold x, not values)
    r = n // 3
    c = int(ci[rp[r] + 1])
    A.set_entry(r, c, 3.25)
    va2 = va.copy()
    va2[rp[r] + 1] = 3.25
    y = np.full(n, np.nan)
    A.matvec_mult(1.0, x, y)
    check_y((rp, ci, va2, n), x, y, 1.0)
    # save / restore: the restored matrix runs the same kernel
    f = str(tmp_path / "m.spx")
    A.save(f)
    B = sx.mat_restore(f)
    assert B.info().unit_windows == 1
    y2 = np.full(n, np.nan)
    B.matvec_mult(1.0, x, y2)
    check_y((rp, ci, va2, n), x, y2, 1.0)
    # the product cut into launches over parts of the rows
    xd = torch.from_numpy(x).cuda()
    yd = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    launched = A.hip_matvec_parts(1.0, xd.data_ptr(), 0.0, yd.data_ptr(), 3, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert launched >= 1
    check_y((rp, ci, va2, n), x, yd.cpu().numpy(), 1.0)
    sx.options_reset()


def test_deterministic_mode_does_not_use_the_windows():
    csr = synth.syn_nlpkkt(12)
    A = tune(csr, dict({"spx.gpu.deterministic": "true"}, **ON))
    assert A.info().unit_windows == 0
    n = csr[3]
    x = synth.random_x(n)
    y = np.full(n, np.nan)
    A.matvec_mult(1.0, x, y)
    check_y(csr, x, y, 1.0)
    sx.options_reset()


@pytest.mark.parametrize("xoff,yoff", [(0, 0), (1, 0), (0, 1), (1, 1), (3, 5)])
def test_device_vectors_at_any_8_byte_alignment(xoff, yoff):
    """spx_hip_matvec_mult takes any double* (reference: spx_matvec_mult takes any vector_t, matvec.c:551-584): x
    and y that start 8 bytes off a 16-byte boundary -- the windows are staged 16 bytes per lane."""
    csr = synth.syn_nlpkkt(14)
    n = csr[3]
    A = tune(csr, dict(ON, **{"spx.preproc.sampling": "none"}))
    assert A.info().unit_windows == 1
    xh = synth.random_x(n)
    xbuf = torch.zeros(n + 8, dtype=torch.float64, device="cuda")
    ybuf = torch.full((n + 8,), float("nan"), dtype=torch.float64, device="cuda")
    xd, yd = xbuf[xoff:xoff + n], ybuf[yoff:yoff + n]
    xd.copy_(torch.from_numpy(xh))
    assert xd.data_ptr() % 16 == 8 * (xoff % 2)
    st = torch.cuda.current_stream().cuda_stream
    A.hip_matvec_mult(0.5, xd.data_ptr(), yd.data_ptr(), st)
    torch.cuda.synchronize()
    check_y(csr, xh, yd.cpu().numpy(), 0.5)
    y0 = synth.random_x(n, seed=5)
    yd.copy_(torch.from_numpy(y0))
    A.hip_matvec_kernel(2.0, xd.data_ptr(), -0.5, yd.data_ptr(), st)
    torch.cuda.synchronize()
    check_y(csr, xh, yd.cpu().numpy(), 2.0, -0.5, y0)
    # nothing was written outside y
    out = ybuf.cpu().numpy()
    assert np.all(np.isnan(out[:yoff])) and np.all(np.isnan(out[yoff + n:]))
    sx.options_reset()
