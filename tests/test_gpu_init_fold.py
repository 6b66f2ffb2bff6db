"""GPU parity of the init pass folded into the launch (spx.gpu.init_fold; InitFold, sparsex_amd/csrc/spmv_device.hpp):
the first workgroups of an adding kernel put beta * y + alpha * diag * x (symmetric kernels with the atomic hand-over;
the reference zeroes y and adds, src/internals/CsxKernels.cpp:105-129) or beta * y (column slices of the general path)
into y, the row-blocks' workgroups wait for them before they add.  Through the C ABI against CSR and the oracle, with
beta != 0, with rows that store themselves (SPX_RB_PRIVATE), repeated launches (the counters must come back to rest)
and vectors 8 bytes off a 16-byte boundary (the init stores go out 16 bytes per lane)."""
import numpy as np
import pytest
import torch

import sparsex_amd as sx
from sparsex_amd import synth
from helpers import tune, oracle_y, check_y, check_vs_oracle

pytestmark = pytest.mark.gpu

FOLD = {"spx.gpu.init_fold": "true", "spx.preproc.sampling": "none"}
SYM = dict(FOLD, **{"spx.matrix.symmetric": "true"})

CASES = [
    ("nd24k-tiles-atomic", lambda: synth.syn_nd24k(0.05), dict(SYM, **{"spx.gpu.sym_spill": "atomic"})),
    ("nd24k-tiles-atomic-w8", lambda: synth.syn_nd24k(0.04), dict(SYM, **{"spx.gpu.sym_spill": "atomic", "spx.gpu.waves": "8"})),
    ("kkt-segments-sx", lambda: synth.syn_nlpkkt(44), dict(SYM, **{"spx.gpu.sym_segments": "true", "spx.gpu.sym_pipeline": "true"})),
    ("kkt-segments-plain", lambda: synth.syn_nlpkkt(36), dict(SYM, **{"spx.gpu.sym_segments": "true", "spx.gpu.sym_pipeline": "false"})),
    ("kkt2f-segments", lambda: synth.syn_kkt2f(40), dict(SYM, **{"spx.gpu.sym_segments": "true"})),
    ("cant-tiles-and-segments", lambda: synth.syn_cant(0.2), dict(SYM, **{"spx.gpu.sym_segments": "true"})),
    ("cant-tiles-atomic", lambda: synth.syn_cant(0.2), dict(SYM, **{"spx.gpu.sym_spill": "atomic"})),
    ("webbase-slices-c2", lambda: synth.syn_webbase(0.2), dict(FOLD, **{"spx.gpu.col_phases": "c2"})),
    ("webbase-slices-c4", lambda: synth.syn_webbase(0.2), dict(FOLD, **{"spx.gpu.col_phases": "c4", "spx.gpu.waves": "8"})),
]


@pytest.mark.parametrize("name,gen,opts", CASES, ids=[c[0] for c in CASES])
def test_mult_and_kernel_with_the_init_pass_folded(name, gen, opts):
    csr = gen()
    n = csr[3]
    A = tune(csr, opts)
    assert A.info().init_fold == 1, "this stream should be able to fold its init pass"
    x = synth.random_x(n)
    for rep in range(3):                       # (the counters are back at rest after every launch)
        y = np.full(n, np.nan)
        A.matvec_mult(0.5, x, y)
        check_y(csr, x, y, 0.5)
    yo, _ = oracle_y(A, x, 0.5)
    check_vs_oracle(csr, x, y, yo, 0.5)
    y0 = synth.random_x(n, seed=7)
    y = y0.copy()
    A.matvec_kernel(1.5, x, -0.25, y)
    check_y(csr, x, y, 1.5, -0.25, y0)
    sx.options_reset()


def test_streams_that_cannot_fold_keep_their_init_kernel():
    # spilled lists (no atomic hand-over), a general stream, the deterministic mode
    for gen, opts in ((lambda: synth.syn_nd24k(0.03), dict(SYM, **{"spx.gpu.sym_spill": "lists"})),
                      (lambda: synth.syn_nlpkkt(16), dict(FOLD)),
                      (lambda: synth.syn_nd24k(0.03), dict(SYM, **{"spx.gpu.deterministic": "true"}))):
        csr = gen()
        A = tune(csr, opts)
        assert A.info().init_fold == 0
        n = csr[3]
        x = synth.random_x(n)
        y = np.full(n, np.nan)
        A.matvec_mult(1.0, x, y)
        check_y(csr, x, y, 1.0)
    sx.options_reset()


def test_auto_measures_and_is_right_either_way(tmp_path):
    for gen, opts in ((lambda: synth.syn_nd24k(0.3), {"spx.matrix.symmetric": "true"}),
                      (lambda: synth.syn_webbase(0.5), {})):
        csr = gen()
        A = tune(csr, opts)
        n = csr[3]
        x = synth.random_x(n)
        y = np.full(n, np.nan)
        A.matvec_mult(0.5, x, y)
        check_y(csr, x, y, 0.5)
        fold = A.info().init_fold
        assert fold in (0, 1)
        # the choice survives save / restore
        f = str(tmp_path / "m.spx")
        A.save(f)
        B = sx.mat_restore(f)
        assert B.info().init_fold == fold
        y2 = np.full(n, np.nan)
        B.matvec_mult(0.5, x, y2)
        check_y(csr, x, y2, 0.5)
    sx.options_reset()


@pytest.mark.parametrize("xoff,yoff", [(1, 0), (0, 1), (3, 5)])
def test_device_vectors_at_any_8_byte_alignment(xoff, yoff):
    csr = synth.syn_nlpkkt(44)
    n = csr[3]
    A = tune(csr, dict(SYM, **{"spx.gpu.sym_segments": "true"}))
    assert A.info().init_fold == 1
    xh = synth.random_x(n)
    xbuf = torch.zeros(n + 8, dtype=torch.float64, device="cuda")
    ybuf = torch.full((n + 8,), float("nan"), dtype=torch.float64, device="cuda")
    xd, yd = xbuf[xoff:xoff + n], ybuf[yoff:yoff + n]
    xd.copy_(torch.from_numpy(xh))
    st = torch.cuda.current_stream().cuda_stream
    A.hip_matvec_mult(0.5, xd.data_ptr(), yd.data_ptr(), st)
    torch.cuda.synchronize()
    check_y(csr, xh, yd.cpu().numpy(), 0.5)
    y0 = synth.random_x(n, seed=5)
    yd.copy_(torch.from_numpy(y0))
    A.hip_matvec_kernel(2.0, xd.data_ptr(), -0.5, yd.data_ptr(), st)
    torch.cuda.synchronize()
    check_y(csr, xh, yd.cpu().numpy(), 2.0, -0.5, y0)
    out = ybuf.cpu().numpy()
    assert np.all(np.isnan(out[:yoff])) and np.all(np.isnan(out[yoff + n:]))
    sx.options_reset()
