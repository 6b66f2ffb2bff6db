"""What a multi-GPU step moves, from the plan objects alone (HISTORY.md section 8; tools/dist_volumes.py prints the
same figures at bench sizes): 2, 4 and 8 ranks of the KKT stand-in, renumbered for the ranks (rcm_owner), every rank
tuning only its rows, the exchange plans built for real -- the ranks are threads of this process joined by an
in-memory transport, so the test needs neither a GPU nor a process group.  A change of the plan that makes the
predicted step worse (a larger halo, a busier peer, entries travelling twice, a round too many) fails here, not in
a document.  The reference partitions rows by nonzeros (include/sparsex/internals/SparseInternal.hpp:131-144) and
finds the rows that are written from outside in CsxBuild.hpp:400-581; the halo of x has no counterpart there."""
import os
import sys
import threading

import numpy as np
import pytest
import scipy.sparse as sp

import sparsex_amd as sx
from sparsex_amd import synth
from sparsex_amd.api import CallbackTransport

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EDGE = 28


class Loopback:
    """All-to-all of uint64 words between the threads of one process."""

    def __init__(self, world):
        self.world = world
        self.box = [None] * world
        self.barrier = threading.Barrier(world)

    def transport(self, rank):
        world = self.world

        def host(send, soff, scnt, recv, roff, rcnt):
            self.box[rank] = (send.copy(), soff.copy(), scnt.copy())
            self.barrier.wait()
            for q in range(world):
                if q == rank:
                    continue
                s, so, sc = self.box[q]
                n = int(rcnt[q])
                assert n == int(sc[rank])
                recv[int(roff[q]):int(roff[q]) + n] = s[int(so[rank]):int(so[rank]) + n]
            self.barrier.wait()

        def device(*args):
            raise RuntimeError("host-only matrices exchange nothing on the device")
        return CallbackTransport(rank, world, host, device)


def renumbered(world):
    sys.path.insert(0, ROOT)
    import bench
    rp, ci, va, n = synth.syn_nlpkkt_rows(EDGE)
    perm = sx.dist_reorder(rp, ci, n, world, sx.SPX_DIST_REORDER_RCM_OWNER, pattern_symmetric=True)
    inv = np.argsort(perm)
    b = sp.csr_matrix((va, ci, rp), shape=(n, n))[inv][:, inv].tocsr()
    b.sort_indices()
    cuts = bench.nnz_balanced_cuts(np.diff(b.indptr), world)
    return b, [int(c) for c in cuts], n


@pytest.mark.parametrize("world", [2, 4, 8])
def test_plans_of_the_renumbered_stencil(world):
    b, cuts, n = renumbered(world)
    # every rank tunes its rows only (the options are the process': one rank after the other)
    mats = []
    for g in range(world):
        lo, hi = cuts[g], cuts[g + 1]
        rl = (b.indptr[lo:hi + 1] - b.indptr[lo]).astype(np.int32)
        cl = b.indices[b.indptr[lo]:b.indptr[hi]].astype(np.int32)
        vl = b.data[b.indptr[lo]:b.indptr[hi]].copy()
        sx.options_reset()
        for k, v in {"spx.rt.host_only": "true", "spx.preproc.sampling": "none", "spx.rt.nr_threads": "2",
                     "spx.rt.row_offset": lo, "spx.rt.global_rows": n, "spx.rt.dist_chunks": 4}.items():
            sx.option_set(k, str(v))
        mats.append(sx.mat_tune(sx.input_load_csr(rl, cl, vl, hi - lo, n)))
    # ... and the plans are built collectively: one thread per rank
    net = Loopback(world)
    errors = []

    def attach(g):
        try:
            mats[g].dist_attach(net.transport(g))
        except Exception as e:           # (a failed thread must not leave the others in the barrier)
            errors.append((g, repr(e)))
            net.barrier.abort()
    threads = [threading.Thread(target=attach, args=(g,)) for g in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors and not any(t.is_alive() for t in threads), errors

    plane = EDGE * EDGE
    received, busiest = [], []
    for g in range(world):
        lo, hi = cuts[g], cuts[g + 1]
        halo = mats[g].dist_halo()
        # exactly the columns outside the own rows that the own rows read, grouped by owner
        c = b[lo:hi].indices
        need = np.unique(c[(c < lo) | (c >= hi)])
        assert np.array_equal(halo["recv_cols"], need)
        owner = np.searchsorted(np.asarray(cuts[1:]), need, side="right")
        assert np.array_equal(np.bincount(owner, minlength=world), halo["recv_cnt"])
        assert np.all((halo["send_rows"] >= lo) & (halo["send_rows"] < hi))
        received.append(int(need.size))
        busiest.append(int(max(halo["recv_cnt"].max(), halo["send_cnt"].max())))
        # the overlapped step: four rounds, every entry in exactly one of them
        rounds = mats[g].dist_rounds()
        assert len(rounds) == 4
        assert sum(int(rd["recv_cnt"].sum()) for rd in rounds) == need.size
        assert sum(int(rd["send_cnt"].sum()) for rd in rounds) == halo["send_rows"].size
    # what is sent is what is received
    assert sum(int(mats[g].dist_halo()["send_rows"].size) for g in range(world)) == sum(received)
    # the shell, not the slice: a rank's halo is a few grid planes of its neighbours (at bench size, edge 240 and
    # eight ranks: 0.7-3.8 MB against slices of 28 MB), and no single link carries more than a few planes either
    slice_rows = max(cuts[g + 1] - cuts[g] for g in range(world))
    assert max(received) <= 14 * plane, (received, plane)
    assert max(busiest) <= 8 * plane, (busiest, plane)
    if world == 2:
        assert max(received) * 4 < slice_rows, (received, slice_rows)
    # ... against the application's own numbering, where a range of state rows reads a whole range of multipliers
    import bench
    rp, ci, va, _ = synth.syn_nlpkkt_rows(EDGE)
    nat_cuts = bench.nnz_balanced_cuts(np.diff(rp), world)
    a = sp.csr_matrix((va, ci, rp), shape=(n, n))
    natural = []
    for g in range(world):
        lo, hi = nat_cuts[g], nat_cuts[g + 1]
        c = a[lo:hi].indices
        natural.append(int(np.unique(c[(c < lo) | (c >= hi)]).size))
    assert sum(received) * (2 if world < 8 else 1) < sum(natural), (received, natural)
    sx.options_reset()
