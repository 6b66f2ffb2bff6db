"""Device-resident vectors (include/sparsex_hip.h) and a conjugate-gradient
solve that never leaves the GPU: the use the reference's vector helpers
(spx_vec_scale_add, spx_vec_mul, ...; src/internals/Vector.cpp:206-394) are
for, here on HBM-resident data."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import sparsex_amd as sx
from sparsex_amd import synth
from helpers import tune
import torch

pytestmark = pytest.mark.gpu


# (sizes around the kernels' chunks: 1024 / 4096 pairs of doubles per workgroup, eight XCD lists, an odd last element)
@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 1000, 2047, 2048, 2049, 8191, 8193, 12345, 16 * 2048 + 1, 1 << 20,
                               (1 << 20) + 2049, 4 * 2048 * 2048 + 5, 27993600])
def test_blas1_against_numpy(n):
    rng = np.random.RandomState(n % 97)
    a, b = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    A, B, Cc = sx.DeviceVector(host=a), sx.DeviceVector(host=b), sx.DeviceVector(n)
    assert np.array_equal(A.download(), a)
    A.scale_add_into(B, Cc, 0.75)                       # c = a + 0.75 b   (one fma or mul+add)
    assert np.allclose(Cc.download(), a + 0.75 * b, rtol=0, atol=2e-16 * 2)
    A.scale_into(Cc, -2.5)
    assert np.array_equal(Cc.download(), -2.5 * a)
    A.copy_into(Cc)
    assert np.array_equal(Cc.download(), a)
    Cc.init(3.25)
    assert np.array_equal(Cc.download(), np.full(n, 3.25))
    d = A.dot(B)
    ref = float(np.dot(a, b))
    assert abs(d - ref) <= 64 * 2.0 ** -53 * float(np.dot(np.abs(a), np.abs(b))) + 1e-300
    assert A.dot(B) == d                                 # fixed reduction order


def test_cg_on_device_matches_host_solver():
    csr = synth.syn_cant(0.05)            # symmetric, strictly diagonally dominant => SPD
    rp, ci, va, n = csr
    M = tune(csr, {"spx.preproc.sampling": "none"})
    a = sp.csr_matrix((va, ci, rp), shape=(n, n))
    xs = np.random.RandomState(3).uniform(-1, 1, n)
    bh = a @ xs
    b = sx.DeviceVector(host=bh)
    x = sx.DeviceVector(n)
    r = sx.DeviceVector(n)
    p = sx.DeviceVector(n)
    ap = sx.DeviceVector(n)
    b.copy_into(r)                        # x0 = 0  =>  r0 = b
    r.copy_into(p)
    rr = r.dot(r)
    rr0 = rr
    its = 0
    while rr > 1e-24 * rr0 and its < 500:
        sx.matvec_kernel_vec(M, 1.0, p, 0.0, ap)        # ap = A p
        alpha = rr / p.dot(ap)
        x.scale_add_into(p, x, alpha)                   # x += alpha p
        r.scale_add_into(ap, r, -alpha)                 # r -= alpha ap
        rr_new = r.dot(r)
        r.scale_add_into(p, p, rr_new / rr)             # p = r + beta p
        rr = rr_new
        its += 1
    xg = x.download()
    assert its < 500
    assert np.linalg.norm(a @ xg - bh) <= 1e-10 * np.linalg.norm(bh)
    xh, info = spla.cg(a, bh, rtol=1e-12, maxiter=500)
    assert info == 0
    assert np.allclose(xg, xh, rtol=1e-8, atol=1e-10)
    assert np.allclose(xg, xs, rtol=1e-8, atol=1e-10)


def test_read_write_probe_reads_every_chunk_and_stores_its_share():
    """spx_hip_probe_read_write (the roof bench.py prints as roofline.measured_mixed_peak): every workgroup reads its
    whole chunk of src and stores what its threads summed to its stretch of dst; bad shapes are refused."""
    n_chunks, chunk, wr = 37, 4096, 144
    h = np.arange(n_chunks * chunk, dtype=np.float64) % 7.0
    src = sx.DeviceVector(host=h)
    dst = sx.DeviceVector(n_chunks * wr)
    dst.init(-1.0)
    src.probe_read_write(dst, chunk, wr)
    torch.cuda.synchronize()
    got = dst.download().reshape(n_chunks, wr)
    # (thread t of a workgroup sums the 16-byte pieces t, t + 256, ... of its chunk and stores that to its slots)
    pairs = h.reshape(n_chunks, chunk // 2, 2).sum(axis=2)                  # x + y of every double2
    per_thread = pairs.reshape(n_chunks, chunk // 512, 256).sum(axis=1)     # thread t: pieces congruent to t mod 256
    assert np.allclose(got, per_thread[:, :wr], rtol=1e-12)
    assert np.isclose(per_thread.sum(), h.sum())
    with pytest.raises(sx.SpxError):
        src.probe_read_write(dst, 1000, 10)            # not a multiple of 2048
    with pytest.raises(sx.SpxError):
        src.probe_read_write(dst, chunk, chunk + 1)    # more written than read
    small = sx.DeviceVector(8)
    with pytest.raises(sx.SpxError):
        src.probe_read_write(small, chunk, wr)         # dst too short
    for v in (src, dst, small):
        v.destroy()
