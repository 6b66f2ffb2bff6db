"""GPU parity of the pipelined read-once kernel (csx_spmv_sx_kernel, spx.gpu.sym_pipeline): through the C ABI
against the oracle decoding the same tuned matrix (the reference's symmetric driver,
src/templates/csx_sym_spmv_tmpl.c:60-106, restated) and against CSR; on stencils whose runs fill passes of their
own and on matrices where only some passes do; 2 / 4 / 8 wavefronts, narrow and wide row-blocks, the beta path,
set-entry, save / restore, unaligned vectors, several partitions."""
import numpy as np
import pytest
import torch

import sparsex_amd as sx
from sparsex_amd import synth
from helpers import tune, oracle_y, check_y, check_vs_oracle

pytestmark = pytest.mark.gpu

ON = {"spx.matrix.symmetric": "true", "spx.gpu.sym_segments": "true", "spx.gpu.sym_pipeline": "true",
      "spx.preproc.sampling": "none"}

CASES = [
    ("kkt44", lambda: synth.syn_nlpkkt(44)),        # runs of 42 segments: passes of their own, partly filled
    ("kkt70", lambda: synth.syn_nlpkkt(70)),        # 68 = one full pass + a tail that joins the mixed ones
    ("kkt30", lambda: synth.syn_nlpkkt(30)),        # runs too short: everything stays on the plain read-once path
    ("kkt2f", lambda: synth.syn_kkt2f(48)),
    ("cant", lambda: synth.syn_cant(0.2)),
]
OPTS = [
    {},
    {"spx.gpu.waves": "2"},
    {"spx.gpu.waves": "4", "spx.rt.nr_threads": "3"},
    {"spx.gpu.waves": "8", "spx.gpu.sym_wide_rows": "2048"},
    {"spx.gpu.sym_wide_rows": "512", "spx.gpu.rowblock_rows": "200"},
    {"spx.gpu.sym_segment_min": "3", "spx.gpu.inline_desc": "false"},
    {"spx.gpu.sym_pure_passes": "false"},           # no passes of their own: only passes that happen to hold one unit
    {"spx.gpu.band_order": "true"},                 # row-blocks uploaded in another order: the plan follows
]


@pytest.mark.parametrize("name,gen", CASES)
@pytest.mark.parametrize("opts", OPTS)
def test_mult_pipelined(name, gen, opts):
    csr = gen()
    A = tune(csr, dict(ON, **opts))
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
    assert inf.sym_segments >= 1
    if name in ("kkt44", "kkt70") and opts.get("spx.gpu.sym_pure_passes") != "false":
        assert inf.sym_pipeline == 1 and inf.sym_pipeline_elems > 0.4 * inf.nnz_stored
    sx.options_reset()


@pytest.mark.parametrize("name,gen", CASES[:4])
def test_kernel_beta_pipelined(name, gen):
    csr = gen()
    A = tune(csr, dict(ON))
    n = csr[3]
    x = synth.random_x(n)
    y0 = synth.random_x(n, seed=7)
    y = y0.copy()
    A.matvec_kernel(1.5, x, -0.25, y)
    check_y(csr, x, y, 1.5, -0.25, y0)
    sx.options_reset()


def test_on_and_off_agree_with_the_oracle():
    csr = synth.syn_nlpkkt(50)
    n = csr[3]
    x = synth.random_x(n)
    ys = {}
    for mode in ("true", "false", "auto"):
        A = tune(csr, dict(ON, **{"spx.gpu.sym_pipeline": mode}))
        y = np.full(n, np.nan)
        A.matvec_mult(1.0, x, y)
        check_y(csr, x, y, 1.0)
        assert A.info().sym_pipeline == {"true": 1, "false": 0}.get(mode, A.info().sym_pipeline)
        ys[mode] = y
    sx.options_reset()


def test_save_restore_and_set_entry(tmp_path):
    csr = synth.syn_nlpkkt(46)
    rp, ci, va, n = csr
    A = tune(csr, dict(ON))
    assert A.info().sym_pipeline == 1
    x = synth.random_x(n)
    # a stored value of the lower triangle changed in HBM is seen by the kernel, in both of its uses
    r = n - n // 5
    k = int(rp[r])
    c = int(ci[k])
    assert c < r
    A.set_entry(r, c, 3.25)
    va2 = va.copy()
    va2[k] = 3.25
    kk = int(rp[c]) + int(np.searchsorted(ci[rp[c]:rp[c + 1]], r))
    assert ci[kk] == r
    va2[kk] = 3.25
    y = np.full(n, np.nan)
    A.matvec_mult(1.0, x, y)
    check_y((rp, ci, va2, n), x, y, 1.0)
    f = str(tmp_path / "m.spx")
    A.save(f)
    B = sx.mat_restore(f)
    assert B.info().sym_pipeline == 1
    y2 = np.full(n, np.nan)
    B.matvec_mult(1.0, x, y2)
    check_y((rp, ci, va2, n), x, y2, 1.0)
    sx.options_reset()


def test_deterministic_mode_does_not_use_the_pipeline():
    csr = synth.syn_nlpkkt(44)
    A = tune(csr, dict(ON, **{"spx.gpu.deterministic": "true"}))
    assert A.info().sym_pipeline == 0
    n = csr[3]
    x = synth.random_x(n)
    y = np.full(n, np.nan)
    A.matvec_mult(1.0, x, y)
    check_y(csr, x, y, 1.0)
    sx.options_reset()


@pytest.mark.parametrize("xoff,yoff", [(1, 0), (0, 1), (3, 5)])
def test_device_vectors_at_any_8_byte_alignment(xoff, yoff):
    """x is read with 16-byte loads at 8-byte granularity (two doubles of a segment at once): vectors that start 8
    bytes off a 16-byte boundary; nothing may be written outside y."""
    csr = synth.syn_nlpkkt(44)
    n = csr[3]
    A = tune(csr, dict(ON))
    assert A.info().sym_pipeline == 1
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
