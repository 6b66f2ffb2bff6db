"""Launch configuration of the SpMV kernel: 2, 4 or 8 wavefronts per workgroup,
pinned with spx.gpu.waves or measured by spx_mat_tune (the default).  Every
configuration computes the same product."""
import numpy as np
import pytest

import sparsex_amd as sx
from sparsex_amd import synth
from helpers import tune, check_y

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("waves", ["2", "4", "8"])
@pytest.mark.parametrize("name,gen,sym", [
    ("cant", lambda: synth.syn_cant(0.1), False),
    ("web", lambda: synth.syn_webbase(0.05), False),
    ("nd24k-sym", lambda: synth.syn_nd24k(0.03), True),
])
def test_pinned_wave_counts_agree(name, gen, sym, waves):
    csr = gen()
    n = csr[3]
    A = tune(csr, {"spx.gpu.waves": waves}, sym=sym)
    assert A.info().waves == int(waves)
    x = synth.random_x(n)
    y = np.full(n, np.nan)
    A.matvec_mult(0.5, x, y)
    check_y(csr, x, y, 0.5)
    y0 = synth.random_x(n, seed=9)
    y = y0.copy()
    A.matvec_kernel(2.0, x, -0.5, y)
    check_y(csr, x, y, 2.0, -0.5, y0)


def test_autotuned_launch_is_one_of_the_built_kernels_and_survives_save(tmp_path):
    csr = synth.syn_cant(0.3)
    n = csr[3]
    A = tune(csr, {})
    w = A.info().waves
    assert w in (2, 4, 8)
    x = synth.random_x(n)
    y = np.zeros(n)
    A.matvec_mult(1.0, x, y)
    check_y(csr, x, y, 1.0)
    f = str(tmp_path / "m.spx")
    A.save(f)
    rb = A.info().n_rowblocks
    A.destroy()
    sx.options_reset()
    B = sx.mat_restore(f)
    assert B.info().waves == w and B.info().n_rowblocks == rb
    y2 = np.zeros(n)
    B.matvec_mult(1.0, x, y2)
    check_y(csr, x, y2, 1.0)
