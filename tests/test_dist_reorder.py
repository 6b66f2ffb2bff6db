"""Partition-aware numbering of a row-partitioned matrix (spx_hip_dist_reorder,
spx.rt.dist_reorder): the permutation is one, the owner form keeps the rows' relative
order inside every rank, the halo of x shrinks on the KKT layout, spx_mat_tune applies it
in front of the nonzero-balanced cut (reference rule SparseInternal.hpp:131-144; its own
reordering is one-process RCM, Rcm.hpp:85-121), and bench.py's ranks generate exactly the
rows of P A P^T that the numbering deals them."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp
import torch.distributed as dist
import torch.multiprocessing as mp

import sparsex_amd as sx
from sparsex_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def halo_sizes(a, cuts):
    out = []
    for g in range(len(cuts) - 1):
        lo, hi = cuts[g], cuts[g + 1]
        c = a[lo:hi].indices
        out.append(int(np.unique(c[(c < lo) | (c >= hi)]).size))
    return out


@pytest.mark.parametrize("world", [2, 4])
def test_owner_order_keeps_relative_order_and_shrinks_the_halo(world):
    sys.path.insert(0, ROOT)
    import bench
    rp, ci, va, n = synth.syn_nlpkkt_rows(14)
    a = sp.csr_matrix((va, ci, rp), shape=(n, n))
    natural = halo_sizes(a, bench.nnz_balanced_cuts(np.diff(rp), world))
    perms = {}
    for name, mode in (("rcm", sx.SPX_DIST_REORDER_RCM), ("rcm_owner", sx.SPX_DIST_REORDER_RCM_OWNER)):
        perm = sx.dist_reorder(rp, ci, n, world, mode, pattern_symmetric=True)
        assert np.array_equal(np.sort(perm), np.arange(n))
        # the same without the promise of a symmetric pattern (A + A^T is built)
        assert np.array_equal(perm, sx.dist_reorder(rp, ci, n, world, mode))
        perms[name] = perm
        inv = np.argsort(perm)
        b = a[inv][:, inv].tocsr()
        cuts = bench.nnz_balanced_cuts(np.diff(b.indptr), world)
        h = halo_sizes(b, cuts)
        # every rank needs less of the others than in the application's numbering, where a
        # range of state rows reads a whole range of multipliers
        assert max(h) < max(natural) and sum(h) < 0.75 * sum(natural), (h, natural)
        if name == "rcm_owner":
            # inside every range the original rows appear in ascending order
            for g in range(world):
                old = inv[cuts[g]:cuts[g + 1]]
                interior = old[8:-8] if old.size > 32 else old     # (the library's own cut may differ by a row or two)
                assert np.all(np.diff(interior) > 0)


def test_one_based_and_argument_checks():
    rp, ci, va, n = synth.syn_cant(0.01)
    L = sx.lib()
    perm = np.empty(n, dtype=np.int32)
    import ctypes as C
    L.spx_hip_dist_reorder.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    rp1, ci1 = (rp + 1).astype(np.int32), (ci + 1).astype(np.int32)
    assert L.spx_hip_dist_reorder(rp1.ctypes.data, ci1.ctypes.data, n, sx.SPX_INDEX_ONE_BASED, 2,
                                  sx.SPX_DIST_REORDER_RCM, 0, perm.ctypes.data) == sx.SPX_SUCCESS
    assert np.array_equal(perm, sx.dist_reorder(rp, ci, n, 2, sx.SPX_DIST_REORDER_RCM))
    old = L.spx_err_get_handler
    for bad in ((n, 99, 2, 1), (n, sx.SPX_INDEX_ZERO_BASED, 0, 1), (n, sx.SPX_INDEX_ZERO_BASED, 2, 7)):
        assert L.spx_hip_dist_reorder(rp.ctypes.data, ci.ctypes.data, bad[0], bad[1], bad[2], bad[3], 0,
                                      perm.ctypes.data) == sx.SPX_FAILURE


@pytest.mark.parametrize("mode", ["rcm", "rcm_owner"])
@pytest.mark.parametrize("symmetric", [False, True])
def test_tune_applies_the_numbering_before_the_cut(tmp_path, mode, symmetric):
    """The whole matrix given to each of two processes (spx.rt.gpu_rank / gpu_world): with
    spx.rt.dist_reorder the tuned matrix is P A P^T, spx_mat_get_perm() returns P, and the
    rows each process owns multiply like the rows of the permuted matrix (host only: the
    saved stream through the numpy decoder)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from stream_decode import Stream
    rp, ci, va, n = synth.syn_nlpkkt_rows(8)
    a = sp.csr_matrix((va, ci, rp), shape=(n, n))
    x = synth.random_x(n)
    covered = np.zeros(n, dtype=bool)
    perms = []
    for rank in range(2):
        sx.options_reset()
        for k, v in {"spx.rt.host_only": "true", "spx.preproc.sampling": "none", "spx.rt.nr_threads": "4",
                     "spx.rt.gpu_rank": rank, "spx.rt.gpu_world": 2, "spx.rt.dist_reorder": mode,
                     "spx.matrix.symmetric": "true" if symmetric else "false"}.items():
            sx.option_set(k, str(v))
        A = sx.mat_tune(sx.input_load_csr(rp, ci, va, n, n))
        perm = A.get_perm()
        assert perm is not None and np.array_equal(np.sort(perm), np.arange(n))
        perms.append(perm)
        inv = np.argsort(perm)
        b = a[inv][:, inv].tocsr()
        info = A.info()
        lo, hi = info.row_lo, info.row_hi
        f = str(tmp_path / ("r%d.spx" % rank))
        A.save(f)
        y = Stream(f).matvec(x)
        want = b @ x
        if symmetric:
            # a symmetric slice also adds into rows in front of its own: compare what it owns after
            # adding the other slice's contribution there (rank 1 -> rows of rank 0)
            covered[lo:hi] = True
            if rank == 0:
                y0 = y.copy()
            else:
                tot = y0 + y
                assert np.allclose(tot, want, rtol=1e-12, atol=1e-13)
        else:
            assert np.allclose(y[lo:hi], want[lo:hi], rtol=1e-12, atol=1e-13)
            covered[lo:hi] = True
        A.destroy()
    assert covered.all() and np.array_equal(perms[0], perms[1])
    sx.options_reset()


def _bench_worker(rank, world, port, symmetric, mode, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import argparse
        import bench
        args = argparse.Namespace(workload="syn-nlpkkt", edge=10, scale=1.0, mtx=None, dist_reorder=mode)
        wl = bench.Workload(args, rank, world, symmetric, dist)
        ret[rank] = (wl.lo, wl.hi, wl.rp, wl.ci, wl.va, wl.n, wl.nnz, wl.lower_local)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("symmetric", [False, True])
@pytest.mark.parametrize("mode", ["rcm", "rcm_owner"])
def test_bench_ranks_generate_their_rows_of_the_renumbered_matrix(symmetric, mode):
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    world = 3
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_bench_worker, args=(world, port, symmetric, mode, ret), nprocs=world, join=True)
    rp, ci, va, n = synth.syn_nlpkkt_rows(10)
    a = sp.csr_matrix((va, ci, rp), shape=(n, n))
    perm = sx.dist_reorder(rp, ci, n, world, {"rcm": sx.SPX_DIST_REORDER_RCM, "rcm_owner": sx.SPX_DIST_REORDER_RCM_OWNER}[mode],
                           pattern_symmetric=True)
    inv = np.argsort(perm)
    b = a[inv][:, inv].tocsr()
    b.sort_indices()
    at = 0
    for r in range(world):
        lo, hi, rpl, cil, val, nn, nnz, lower = ret[r]
        assert lo == at and nn == n and nnz == rp[-1]
        at = hi
        s = b[lo:hi]
        assert np.array_equal(s.indptr, rpl) and np.array_equal(s.indices, cil) and np.array_equal(s.data, val)
        if symmetric:
            coo = s.tocoo()
            assert lower == int((coo.col < coo.row + lo).sum())
    assert at == n
