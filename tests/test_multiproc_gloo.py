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
