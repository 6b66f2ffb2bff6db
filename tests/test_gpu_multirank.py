"""Several ranks on the GPU box (which has ONE MI355X): fresh child processes
share device 0 and talk over gloo (SPX_BENCH_BACKEND=gloo), so everything of
the row-partitioned path runs on the GPU except the transport -- row-slice
inputs, per-rank tuning, the symmetric init / tile / mirror kernels of a slice,
pack -> pairwise exchange -> ordered add, the y hand-round -- with bench.py's
per-rank parity gate against the CSR product of the rank's own rows.  The rank
boundaries cut through coupled rows (a 27-point stencil, dense 8x8 blocks), so
the symmetric exchange carries real sums.  The children are started before they
touch the GPU; nothing is re-exec'ed."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    """A TCP port nobody listens on right now (bound to port 0, read back, released): fixed
    numbers per case collided across back-to-back runs of the suite (EADDRINUSE)."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def run_bench(world, extra, port=None, timeout=900, host_threads=2):
    port = port or free_port()
    env = dict(os.environ)
    env.update({"SPX_BENCH_BACKEND": "gloo", "MASTER_ADDR": "127.0.0.1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", str(world), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--host-threads",
           str(host_threads)] + extra
    if world == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
               "--no-cpu-baseline", "--no-configs", "--host-threads", "2"] + extra
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name,extra,tiles", [
    # (a general-path run on several ranks also takes the symmetric path, in the same invocation)
    ("nlpkkt", ["--edge", "28", "--dist-reorder", "none"], False),
    ("nlpkkt-sym", ["--edge", "28", "--symmetric", "--dist-reorder", "none"], False),
    ("nlpkkt-sym-segments", ["--edge", "28", "--symmetric", "--opt", "spx.gpu.sym_segments=true"], False),
    # runs long enough for passes of their own: the pipelined read-once kernel on a rank's slice
    ("nlpkkt-sym-pipelined", ["--edge", "48", "--symmetric", "--dist-reorder", "none", "--opt", "spx.gpu.sym_segments=true",
                              "--opt", "spx.gpu.sym_pipeline=true"], False),
    ("kkt2f-sym", ["--workload", "syn-kkt2f", "--edge", "28", "--symmetric"], False),
    ("kkt2f-sym-segments", ["--workload", "syn-kkt2f", "--edge", "28", "--symmetric",
                            "--opt", "spx.gpu.sym_segments=true"], False),
    # big enough per rank (>= 16 M nonzeros in its triangle) for the library to choose the segments itself
    ("kkt2f-sym-auto", ["--workload", "syn-kkt2f", "--edge", "100", "--symmetric", "--host-threads", "16"], False),
    ("nd24k-sym", ["--workload", "syn-nd24k", "--scale", "0.15", "--symmetric"], True),
    ("nd24k-sym-atomic", ["--workload", "syn-nd24k", "--scale", "0.15", "--symmetric",
                          "--opt", "spx.gpu.sym_spill=atomic"], True),
    ("webbase", ["--workload", "syn-webbase", "--scale", "0.1"], False),
    # the unknowns renumbered for the ranks (spx_hip_dist_reorder) before the rows are dealt
    ("nlpkkt-rcm", ["--edge", "28", "--dist-reorder", "rcm"], False),
    ("nlpkkt-rcm_owner", ["--edge", "28"], False),                       # (the default on several ranks)
    ("nd24k-rcm_owner", ["--workload", "syn-nd24k", "--scale", "0.15", "--dist-reorder", "rcm_owner"], True),
], ids=["nlpkkt", "nlpkkt-sym", "nlpkkt-sym-segments", "nlpkkt-sym-pipelined", "kkt2f-sym", "kkt2f-sym-segments", "kkt2f-sym-auto",
        "nd24k-sym", "nd24k-sym-atomic", "webbase", "nlpkkt-rcm", "nlpkkt-rcm_owner", "nd24k-rcm_owner"])
def test_ranks_share_one_gpu(world, name, extra, tiles):
    out = run_bench(world, extra)
    check_line(out, world, name)


def check_line(out, world, name):
    assert out["n_gpus"] == world and out["scaling"] == "strong" and out["value"] > 0
    paths = [out] + ([out["symmetric"]] if "symmetric" in out else [])
    assert ("symmetric" in out) == (not out["config"]["symmetric_path"] and name not in ("webbase", "nlpkkt-only"))
    if name in ("kkt2f-sym-segments", "kkt2f-sym-auto", "nlpkkt-sym-segments"):
        assert "symseg" in out["roofline"]["kernel"]
    if name == "nlpkkt-sym-pipelined":
        assert "csx_spmv_sx_kernel" in out["roofline"]["kernel"]
    for res in paths:
        assert res["parity"]["max_err_over_fp64_bound"] <= 1.0
        ranks = res["ranks"]
        assert [r["rank"] for r in ranks] == list(range(world))
        # the ranks' rows tile the matrix in order, with comparable nonzero counts (of what they store)
        assert ranks[0]["rows"][0] == 0 and ranks[-1]["rows"][1] == res["config"]["nrows"]
        assert all(ranks[i]["rows"][1] == ranks[i + 1]["rows"][0] for i in range(world - 1))
        assert sum(r["nnz"] for r in ranks) == res["config"]["nnz"]
        assert max(r["balance_nnz"] for r in ranks) < 1.5 * min(r["balance_nnz"] for r in ranks)
        # `value` is a full iteration step: it includes the halo exchange of y (and the conflict exchange);
        # the halo is at most what the hand-round of whole slices would bring
        col = res["collective"]
        assert res["config"]["collective_in_value"].startswith("included")
        assert col["full_step_gflops"] == pytest.approx(res["value"], rel=5e-3, abs=0.02)
        assert col["full_step_ms"] >= col["owned_rows_only_ms"] * 0.8 and col["kernels_only_gflops"] > 0
        assert col["gather_y_step_ms"] > 0
        assert col["halo_bytes_received_per_rank"] <= col["y_handround_bytes_received_per_rank"]   # (rank 0's)
        assert all(8 * r["halo_entries_received"] <= 8 * (res["config"]["nrows"] - (r["rows"][1] - r["rows"][0])) for r in ranks)
        assert sum(r["halo_entries_received"] for r in ranks) == sum(r["halo_entries_sent"] for r in ranks) > 0
        sym = res["config"]["symmetric_path"]
        sent = [r["conflict_rows_sent"] for r in ranks]
        if sym:
            # real cross-rank coupling: every rank but the first adds into rows of the ranks in
            # front of it, and what travels is less than an n-long all-reduce would move
            assert sent[0] == 0 and all(0 < s < res["config"]["nrows"] // 2 for s in sent[1:])
            assert sum(r["conflict_entries_received"] for r in ranks) == sum(sent)
        else:
            assert sent == [0] * world


@pytest.mark.gpu
def test_contract_matrix_on_eight_ranks_renumbered():
    """The same eight ranks after spx_hip_dist_reorder(rcm_owner): rank 0 computes the numbering from the
    pattern of all 769 M nonzeros and broadcasts it, every rank generates the rows it is dealt of
    P A P^T.  A rank now reads a thin shell of its neighbours' unknowns instead of a whole slice."""
    world, N = 8, 240
    n = 2 * N ** 3 + 6 * N ** 2
    out = run_bench(world, ["--edge", str(N), "--no-configs", "--dist-reorder", "rcm_owner"], timeout=2400, host_threads=2)
    check_line(out, world, "nlpkkt-only")
    assert out["config"]["dist_reorder"] == "rcm_owner" and out["config"]["nnz"] == 768977264
    ranks, col = out["ranks"], out["collective"]
    assert max(r["halo_entries_received"] for r in ranks) < 0.3 * n / world          # natural order: ~1.0 x n / world
    assert sum(r["halo_entries_received"] for r in ranks) < 0.2 * n
    assert col["overlap_rounds"] >= 2 and all(r["overlap_parts"] >= 2 for r in ranks)


@pytest.mark.gpu
@pytest.mark.parametrize("symmetric", [False, True], ids=["nlpkkt-e240-8", "nlpkkt-e240-sym-8"])
def test_contract_matrix_on_eight_ranks(symmetric):
    """BASELINE config 5 at its real size through the real plan: syn-nlpkkt at grid edge 240
    (27 993 600 rows, 769 M nonzeros) row-partitioned over EIGHT ranks with the reference's rule
    (SparseInternal.hpp:131-144; symmetric: by stored nonzeros, the local buffers of
    src/api/matvec.c:302-318 become the conflict rows of CsxBuild.hpp:400-581, added in the order of
    Vector.cpp:291-299).  Eight fresh processes share the one GPU of this box; each generates and
    tunes only its slice (96 M nonzeros, 0.77 GB of values), gloo carries the exchange.  bench.py
    gates every rank against the CSR product of its own rows for SPX_DIST_OWNED_ROWS, then with
    SPX_DIST_GATHER_Y (all ranks must hold the same y), then with SPX_DIST_HALO_X (own rows again,
    and the halo entries against the gathered y).  The byte counts are those of HISTORY.md section 8."""
    world, N = 8, 240
    n, P = 2 * N ** 3 + 6 * N ** 2, N ** 3 + 6 * N ** 2
    out = run_bench(world, ["--edge", str(N), "--no-configs", "--dist-reorder", "none"] + (["--symmetric"] if symmetric else []),
                    timeout=2400, host_threads=2)
    check_line(out, world, "nlpkkt-sym" if symmetric else "nlpkkt-only")
    assert out["config"]["nrows"] == n and out["config"]["nnz"] == 768977264
    ranks, col = out["ranks"], out["collective"]
    rows = [r["rows"][1] - r["rows"][0] for r in ranks]
    if not symmetric:
        # equal nonzeros = about equal rows (the rank that holds the 345 600 two-entry control rows has
        # a tenth more): 28 MB slices, 7 x 28 MB per rank in the hand-round
        assert all(abs(k - n / world) < 0.12 * n / world for k in rows)
        assert col["y_handround_bytes_received_per_rank"] == 8 * (n - rows[0])
        # in this order of the unknowns a rank of state rows reads a whole slice of multipliers
        # (and the other way round): the halo is about one slice, an eighth of the hand-round
        assert all(0.85 * n / world < r["halo_entries_received"] < 1.25 * n / world for r in ranks)
    else:
        # dealt by STORED nonzeros: rank 0 holds every state and control row (diagonal only) and the
        # first multiplier rows; every other rank adds into 1.93 M state rows (its conflict rows)
        assert ranks[0]["rows"][0] == 0 and ranks[0]["rows"][1] > P
        for r in ranks[1:]:
            assert 1.85e6 < r["conflict_rows_sent"] < 2.0e6
            assert r["halo_entries_received"] == r["conflict_rows_sent"]      # the same columns, read as x
        assert ranks[0]["conflict_entries_received"] == sum(r["conflict_rows_sent"] for r in ranks[1:])


@pytest.mark.gpu
def test_one_rank_runs_the_same_workload():
    """N = 1 is the same matrix on one GPU (what makes the N-axis a strong-scaling curve)."""
    one = run_bench(1, ["--edge", "28"])
    two = run_bench(2, ["--edge", "28"])                      # (renumbered for two ranks: P A P^T, the same operator)
    assert one["config"]["nnz"] == two["config"]["nnz"] and one["config"]["nrows"] == two["config"]["nrows"]
    assert two["config"]["workload"].startswith(one["config"]["workload"]) and two["config"]["dist_reorder"] == "rcm_owner"
    assert one["scaling"] == two["scaling"] == "strong"


@pytest.mark.gpu
def test_rccl_transport_single_rank():
    """The built-in transport on the one GPU of this box: librccl is loaded on demand, a
    communicator of one rank is created from a unique id, a matrix attaches to it and
    spx_hip_matvec_dist runs (nothing to exchange with one rank).  The pairwise exchange
    itself needs several GPUs (RCCL refuses two ranks on one device)."""
    import numpy as np
    import torch
    import sparsex_amd as sx
    from sparsex_amd import synth
    from helpers import tune, check_y
    uid = sx.rccl_unique_id()
    assert len(uid) == 128
    t = sx.RcclTransport(uid, 0, 1)
    for sym in (False, True):
        csr = synth.syn_nd24k(0.03)
        n = csr[3]
        A = tune(csr, {}, sym=sym)
        A.dist_attach(t)
        plan = A.dist_plan()
        assert plan["world"] == 1 and list(plan["row_lo"]) == [0] and list(plan["row_hi"]) == [n]
        assert not plan["any_exchange"] and plan["send_rows"].size == 0
        xh = synth.random_x(n)
        x = torch.from_numpy(xh).cuda()
        y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
        A.hip_matvec_dist(0.5, x.data_ptr(), 0.0, y.data_ptr(), sx.SPX_DIST_GATHER_Y,
                          torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        check_y(csr, xh, y.cpu().numpy(), 0.5)
        A.destroy()
    t.destroy()


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """`python3 bench.py --gpus 2` WITHOUT a launcher -- the way the driver starts the one-GPU run -- starts the two
    ranks itself (a child torch.distributed.run, before the parent touches a GPU), passes rank 0's line through and
    leaves with the child's code (tests/test_bench_launch.py holds the mechanics on the CPU).  Here the ranks do the
    real step on the one GPU of this box over gloo; with RCCL the same line would carry ranks_seen_by_rccl = 2."""
    env = dict(os.environ)
    env.update({"SPX_BENCH_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--host-threads", "2", "--edge", "28"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    check_line(out, 2, "nlpkkt")
    assert "ranks_seen_by_rccl" in out and out["ranks_seen_by_rccl"] is None      # (gloo carried the exchange)


@pytest.mark.gpu
def test_rccl_transport_counts_its_ranks():
    import sparsex_amd as sx
    t = sx.RcclTransport(sx.rccl_unique_id(), 0, 1)
    assert t.rccl_ranks() == 1
    t.destroy()
