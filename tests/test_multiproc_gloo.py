"""N > 1 processes on CPU (gloo, world size 2): each rank owns its slice of
the nnz-balanced row partitions (spx.rt.gpu_rank / gpu_world), exactly as
bench.py does on GPUs.  The arithmetic here is done by the oracle on the
partitions the rank exported (no GPU in this test); what is covered is the
ownership logic, the block-diagonal weak-scaling workload and the collective
composition: no collective on the general path (row slices are disjoint), an
all-reduce of the per-rank partial vectors on the symmetric path."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, symmetric, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sparsex_amd as sx
    from sparsex_amd import synth
    from oracle import pyoracle
    import bench
    base = synth.syn_cant(0.03)
    csr = bench.make_workload("syn-cant", 0.03, copies=world)
    rp, ci, va, n = csr
    assert n == base[3] * world and rp[-1] == base[0][-1] * world
    sx.options_reset()
    for k, v in {"spx.rt.host_only": "true", "spx.preproc.sampling": "none",
                 "spx.rt.nr_threads": 2 * world, "spx.rt.gpu_rank": rank,
                 "spx.rt.gpu_world": world,
                 "spx.matrix.symmetric": "true" if symmetric else "false"}.items():
        sx.option_set(k, str(v))
    A = sx.mat_tune(sx.input_load_csr(rp, ci, va, n, n))
    inf = A.info()
    assert (inf.first_partition, inf.last_partition) == (2 * rank, 2 * rank + 2)
    x = synth.random_x(n)
    ex = [A.export_csx(p) for p in range(inf.first_partition, inf.last_partition)]
    y_part = pyoracle.csx_matvec(pyoracle.Partitions(ex, symmetric), x, n, 0.5)
    if not symmetric:
        # rows outside [row_lo, row_hi) are untouched by this rank
        assert not y_part[:inf.row_lo].any() and not y_part[inf.row_hi:].any()
        bounds = [None] * world
        dist.all_gather_object(bounds, (inf.row_lo, inf.row_hi))
        assert bounds[0][0] == 0 and bounds[-1][1] == n
        assert all(bounds[i][1] == bounds[i + 1][0] for i in range(world - 1))
    t = torch.from_numpy(y_part.copy())
    dist.all_reduce(t)                    # symmetric: the real exchange; general: assembles y
    y = t.numpy()
    yc = 0.5 * pyoracle.csr_matvec(rp, ci, va, x)
    ok = pyoracle.vec_compare(yc, y) == 0
    ret[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("symmetric", [False, True])
def test_two_ranks_gloo(symmetric):
    world = 2
    port = 29600 + (os.getpid() % 200) + (50 if symmetric else 0)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, symmetric, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world))


# ---- the exchange plan of row-sliced matrices (BASELINE config 4's layout) --------------

def _slice_worker(rank, world, port, symmetric, gen, tmpdir, ret, parts=None):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import scipy.sparse as sp
    import sparsex_amd as sx
    from sparsex_amd import synth
    from sparsex_amd.dist_torch import torch_transport
    from stream_decode import Stream
    from test_row_slices import nnz_balanced_bounds
    try:
        if gen == "nlpkkt":
            rp, ci, va, n = synth.syn_nlpkkt_rows(9)
        elif gen == "kkt2f":
            rp, ci, va, n = synth.syn_kkt2f_rows(9)
        else:
            rp, ci, va, n = synth.syn_nd24k(0.02)
        a = sp.csr_matrix((va, ci, rp), shape=(n, n))
        cuts = nnz_balanced_bounds(np.diff(rp), world)
        lo, hi = cuts[rank], cuts[rank + 1]
        # every rank holds ONLY its rows (the whole matrix above is the checker's)
        rl = (rp[lo:hi + 1] - rp[lo]).astype(np.int32)
        cl, vl = ci[rp[lo]:rp[hi]].copy(), va[rp[lo]:rp[hi]].copy()
        sx.options_reset()
        for k, v in {"spx.rt.host_only": "true", "spx.preproc.sampling": "none",
                     "spx.rt.nr_threads": "2", "spx.rt.row_offset": lo, "spx.rt.global_rows": n,
                     "spx.rt.dist_chunks": parts[rank] if parts else 4,
                     "spx.matrix.symmetric": "true" if symmetric else "false"}.items():
            sx.option_set(k, str(v))
        A = sx.mat_tune(sx.input_load_csr(rl, cl, vl, hi - lo, n))
        A.dist_attach(torch_transport(rank, world))
        plan = A.dist_plan()
        assert plan["rank"] == rank and plan["world"] == world
        assert list(plan["row_lo"]) == cuts[:-1] and list(plan["row_hi"]) == cuts[1:]
        # the conflict rows are exactly the columns in front of the slice that its
        # lower triangle touches (the reference's map, CsxBuild.hpp:432-451)
        if symmetric:
            sl = a[lo:hi].tocoo()
            want = np.unique(sl.col[sl.col < lo])
        else:
            want = np.zeros(0, dtype=np.int64)
        assert np.array_equal(plan["send_rows"], want), (plan["send_rows"][:10], want[:10])
        assert plan["any_exchange"] == (symmetric and world > 1)
        # the local product, lane by lane from the saved stream (numpy decoder)
        f = os.path.join(tmpdir, "r%d.spx" % rank)
        A.save(f)
        x = synth.random_x(n)
        y = Stream(f).matvec(x)
        # pack -> pairwise exchange -> add in the plan's fixed order
        send = torch.from_numpy(y[plan["send_rows"]].copy())
        recv = torch.empty(plan["n_recv"], dtype=torch.float64)
        dist.all_to_all_single(recv, send, [int(v) for v in plan["recv_cnt"]],
                               [int(v) for v in plan["send_cnt"]])
        recv = recv.numpy()
        for t, r in enumerate(plan["fix_rows"]):
            y[r] += recv[plan["fix_pos"][plan["fix_ptr"][t]:plan["fix_ptr"][t + 1]]].sum()
        want_y = (a @ x)[lo:hi]
        ok = np.allclose(y[lo:hi], want_y, rtol=1e-12, atol=1e-14)
        # the halo of x (SPX_DIST_HALO_X): exactly the columns outside the slice that its stored
        # nonzeros reach (general: all of them; symmetric: the lower triangle's), by owner; the
        # owners hold the matching lists of rows to pack, and after the exchange this rank has
        # every entry of the full product that its own rows read as x
        halo = A.dist_halo()
        sl = a[lo:hi].tocoo()
        outside = sl.col[(sl.col < lo) | ((sl.col >= hi) & (not symmetric))]
        assert np.array_equal(halo["recv_cols"], np.unique(outside)), (halo["recv_cols"][:8], np.unique(outside)[:8])
        owner = np.searchsorted(np.asarray(cuts[1:]), halo["recv_cols"], side="right")
        assert np.array_equal(np.bincount(owner, minlength=world), halo["recv_cnt"])
        assert np.all((halo["send_rows"] >= lo) & (halo["send_rows"] < hi))
        hs = torch.from_numpy(y[halo["send_rows"]].copy())
        hr = torch.empty(int(halo["recv_cnt"].sum()), dtype=torch.float64)
        dist.all_to_all_single(hr, hs, [int(v) for v in halo["recv_cnt"]], [int(v) for v in halo["send_cnt"]])
        y[halo["recv_cols"]] = hr.numpy()
        ok = ok and np.allclose(y[halo["recv_cols"]], (a @ x)[halo["recv_cols"]], rtol=1e-12, atol=1e-14)
        # the overlapped step (SPX_DIST_OVERLAP): the same entries in rounds -- round r carries what
        # lies in part r of the owner's rows (host-only matrices are cut into equal row parts); every
        # entry travels exactly once, and after the last round the halo is what the one-shot exchange brought
        rounds = A.dist_rounds()
        if not symmetric:
            assert len(rounds) == (min(64, max(parts)) if parts else 4)
            y2 = y.copy()
            y2[halo["recv_cols"]] = np.nan
            sendbuf = y[halo["send_rows"]].copy()
            recvbuf = np.full(int(halo["recv_cnt"].sum()), np.nan)
            seen_s, seen_r = np.zeros(sendbuf.size, dtype=int), np.zeros(recvbuf.size, dtype=int)
            for rd in rounds:
                spos = np.concatenate([np.arange(o, o + c) for o, c in zip(rd["send_off"], rd["send_cnt"])]).astype(np.int64)
                rpos = np.concatenate([np.arange(o, o + c) for o, c in zip(rd["recv_off"], rd["recv_cnt"])]).astype(np.int64)
                seen_s[spos] += 1
                seen_r[rpos] += 1
                got = torch.empty(rpos.size, dtype=torch.float64)
                dist.all_to_all_single(got, torch.from_numpy(sendbuf[spos].copy()), [int(v) for v in rd["recv_cnt"]],
                                       [int(v) for v in rd["send_cnt"]])
                recvbuf[rpos] = got.numpy()
                y2[halo["recv_cols"][rpos]] = recvbuf[rpos]
            assert np.all(seen_s == 1) and np.all(seen_r == 1)
            ok = ok and np.array_equal(y2[halo["recv_cols"]], y[halo["recv_cols"]])
        # ... which is less than the slices of y handed round (general path: far less on a banded matrix)
        ret["halo%d" % rank] = (int(halo["recv_cols"].size), n - (hi - lo))
        # bytes that travel: the conflict entries, not n doubles
        ret[rank] = (bool(ok), int(plan["send_rows"].size), n)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("gen", ["nlpkkt", "kkt2f", "nd24k"])
@pytest.mark.parametrize("symmetric", [False, True])
@pytest.mark.parametrize("world", [2, 3])
def test_row_slices_exchange_plan_gloo(tmp_path, world, symmetric, gen):
    port = 29850 + (os.getpid() % 100) + 7 * world + (3 if symmetric else 0) + {"nlpkkt": 0, "kkt2f": 80, "nd24k": 40}[gen]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_slice_worker, args=(world, port, symmetric, gen, str(tmp_path), ret), nprocs=world, join=True)
    assert all(ret[r][0] for r in range(world))
    assert all(ret["halo%d" % r][0] <= ret["halo%d" % r][1] for r in range(world))
    if gen in ("kkt2f", "nd24k"):          # banded: the halo is a part of the other ranks' slices only
        assert all(ret["halo%d" % r][0] < ret["halo%d" % r][1] for r in range(world))
    if symmetric:
        # rank 0 sends nothing; the others send far less than an n-long all-reduce would move
        # (banded matrices.  In the KKT layout of syn-nlpkkt the stored triangle is the multiplier
        # rows, whose mirror image lands on the state rows -- a band of THAT block, owned by the
        # first ranks: still never more than the rows in front of the sender)
        limit = (lambda r: ret[r][2]) if gen == "nlpkkt" else (lambda r: ret[r][2] // 2)
        assert ret[0][1] == 0 and all(0 < ret[r][1] < limit(r) for r in range(1, world))


def test_rounds_with_different_part_counts_gloo(tmp_path):
    """The ranks may cut their products into different numbers of parts (a rank with few row-blocks
    gets fewer): everybody takes part in as many rounds as the rank with the most parts, and every
    halo entry still travels exactly once (checked inside the worker)."""
    world = 3
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_slice_worker, args=(world, port, False, "nlpkkt", str(tmp_path), ret, [2, 5, 3]), nprocs=world, join=True)
    assert all(ret[r][0] for r in range(world))


def test_rounds_when_one_rank_does_not_cut_and_one_asks_for_too_many_gloo(tmp_path):
    """A rank configured with ONE part takes part in the (collective) planning all the same -- it used to skip
    the exchange of the part bounds and leave the others waiting -- and a rank that asks for more parts
    than the plan has room for gets the 64 the plan allows, not a truncated list."""
    world = 3
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_slice_worker, args=(world, port, False, "nlpkkt", str(tmp_path), ret, [1, 200, 3]), nprocs=world, join=True)
    assert all(ret[r][0] for r in range(world))
