"""The randomised matrices and options of test_stream_random.py through the
kernels on the GPU: product against CSR (reference criterion + fp64 bound)."""
import numpy as np
import pytest

from sparsex_amd import synth
from helpers import tune, check_y
from test_stream_random import random_matrix, random_options, random_sym_options

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(40))
def test_general_random_gpu(seed):
    csr, m = random_matrix(seed, symmetric=False)
    n = csr[3]
    o = random_options(seed)
    o["spx.gpu.waves"] = str([0, 2, 4, 8][seed % 4])
    A = tune(csr, o)
    x = synth.random_x(n)
    y = np.full(n, np.nan)
    A.matvec_mult(0.5, x, y)
    check_y(csr, x, y, 0.5)
    y0 = synth.random_x(n, seed=seed + 1)
    y = y0.copy()
    A.matvec_kernel(-1.0, x, 0.25, y)
    check_y(csr, x, y, -1.0, 0.25, y0)


@pytest.mark.parametrize("seed", range(40, 90))
def test_symmetric_random_gpu(seed):
    csr, m = random_matrix(seed, symmetric=True)
    n = csr[3]
    o = random_sym_options(seed, random_options(seed))
    o["spx.gpu.waves"] = str([0, 2, 4, 8][seed % 4])
    A = tune(csr, o, sym=True)
    x = synth.random_x(n)
    y = np.full(n, np.nan)
    A.matvec_mult(0.5, x, y)
    check_y(csr, x, y, 0.5)
    y0 = synth.random_x(n, seed=seed + 1)
    y = y0.copy()
    A.matvec_kernel(2.0, x, -0.5, y)
    check_y(csr, x, y, 2.0, -0.5, y0)
