"""The device entry points only enqueue kernels on the stream they are given,
so solver loops can be captured into a hipGraph (stream capture through
torch.cuda.CUDAGraph) and replayed with one launch."""
import numpy as np
import pytest
import scipy.sparse as sp

import sparsex_amd as sx
from sparsex_amd import synth
from helpers import tune, check_y

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sym,gen,opts", [
    (False, lambda: synth.syn_cant(0.05), {}),
    (True, lambda: synth.syn_cant(0.05), {}),
    # the pipelined read-once kernel (init kernel + csx_spmv_sx_kernel per product) inside a captured graph
    (True, lambda: synth.syn_nlpkkt(44), {"spx.gpu.sym_segments": "true", "spx.gpu.sym_pipeline": "true"}),
], ids=["general", "symmetric", "symmetric-pipelined"])
def test_captured_spmv_replays(sym, gen, opts):
    import torch
    csr = gen()
    rp, ci, va, n = csr
    A = tune(csr, dict({"spx.preproc.sampling": "none"}, **opts), sym=sym)
    x = torch.from_numpy(synth.random_x(n)).cuda()
    y = torch.zeros(n, dtype=torch.float64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    A.hip_matvec_mult(0.5, x.data_ptr(), y.data_ptr(), s)          # warm-up outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cap = torch.cuda.current_stream().cuda_stream
        for _ in range(4):                                          # y <- 0.5 A x + 0.25 y, four times
            A.hip_matvec_kernel(0.5, x.data_ptr(), 0.25, y.data_ptr(), cap)
    a = sp.csr_matrix((va, ci, rp), shape=(n, n))
    xh = x.cpu().numpy()
    for rep in range(3):                                            # new inputs, same graph
        y.fill_(float(rep))
        g.replay()
        torch.cuda.synchronize()
        ref = np.full(n, float(rep))
        for _ in range(4):
            ref = 0.5 * (a @ xh) + 0.25 * ref
        assert np.allclose(y.cpu().numpy(), ref, rtol=1e-12, atol=1e-13)


def test_captured_richardson_iteration_on_device_vectors():
    import torch
    csr = synth.syn_cant(0.03)           # strictly diagonally dominant
    rp, ci, va, n = csr
    a = sp.csr_matrix((va, ci, rp), shape=(n, n))
    d = a.diagonal()
    omega = 0.9 / np.abs(a).sum(axis=1).max()
    A = tune(csr, {"spx.preproc.sampling": "none"})
    bh = a @ np.random.RandomState(1).uniform(-1, 1, n)
    b, x, r = sx.DeviceVector(host=bh), sx.DeviceVector(n), sx.DeviceVector(n)
    x.init(0.0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cap = torch.cuda.current_stream().cuda_stream
        for _ in range(10):
            b.copy_into(r, cap)
            sx.matvec_kernel_vec(A, -1.0, x, 1.0, r, cap)          # r = b - A x
            x.scale_add_into(r, x, omega, cap)                     # x += omega r
    xr = np.zeros(n)
    for rep in range(5):
        g.replay()
        for _ in range(10):
            xr = xr + omega * (bh - a @ xr)
    torch.cuda.synchronize()
    assert np.allclose(x.download(), xr, rtol=1e-11, atol=1e-13)
    assert np.linalg.norm(bh - a @ xr) < np.linalg.norm(bh)        # it does converge
