#!/usr/bin/env python3
"""Option ablations on the GPU: tunes a workload under several option sets and prints one line
each -- time per SpMV (HIP events on the launch stream, median of 5 batches), index bytes per
stored nonzero, tune and emit seconds, row-blocks, kernel choice -- after gating the product
against CSR.  Output is markdown (a table row per line), so that it can be committed as it is.

usage: tools/abl.py <workload> [--edge N] [--scale S] [--symmetric] [--steps K] [--threads T]
                    [--csv] SET [SET ...]
       SET = name:opt=value,opt=value,...   ("default:" = no options)
workloads: syn-cant syn-nd24k syn-webbase syn-nlpkkt syn-kkt2f syn-bandrandom
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("--edge", type=int, default=120)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--symmetric", action="store_true")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--header", action="store_true")
    ap.add_argument("sets", nargs="+")
    args = ap.parse_args()

    import torch
    import scipy.sparse as sp
    import sparsex_amd as sx
    from sparsex_amd import synth
    import bench

    torch.cuda.set_device(0)
    if args.workload in bench.SLICED:
        csr = synth._rows(bench.SLICED[args.workload], args.edge, 0, None, None, synth.SEED_BASE + 4)
        label = "%s e%d" % (args.workload, args.edge)
    else:
        csr = synth.WORKLOADS[args.workload](args.scale)
        label = args.workload
    rp, ci, va, n = csr
    nnz = int(rp[-1])
    a = sp.csr_matrix((va, ci, rp), shape=(n, n))
    xh = synth.random_x(n)
    yc = bench.ALPHA * (a @ xh)
    bound = 64.0 * 2.0 ** -53 * bench.ALPHA * (abs(a) @ np.abs(xh)) + 1e-300
    x = torch.from_numpy(xh).cuda()
    y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    if args.header:
        print("| workload | path | options | us / SpMV | GFLOP/s | index B/nnz | stored nnz | row-blocks | waves | "
              "tune s | emit+upload s | max err / bound |")
        print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for st in args.sets:
        name, _, body = st.partition(":")
        opts = {"spx.rt.nr_threads": args.threads, "spx.rt.keep_encoded": "false",
                "spx.matrix.symmetric": "true" if args.symmetric else "false"}
        for kv in filter(None, body.split(",")):
            k, v = kv.split("=", 1)
            opts[k] = v
        A = bench.tune(csr, opts)
        info = A.info()
        st_ = torch.cuda.current_stream().cuda_stream
        y.fill_(float("nan"))
        A.hip_matvec_mult(bench.ALPHA, x.data_ptr(), y.data_ptr(), st_)
        torch.cuda.synchronize()
        err = float(np.max(np.abs(y.cpu().numpy() - yc) / bound))
        ablation = os.environ.get("SPX_BENCH_ABLATION") == "1"
        assert err <= 1.0 or ablation, "parity gate failed: %g" % err
        if ablation:             # (a build that computes wrong results on purpose: the row says so)
            name = "INVALID (ablation build, err / bound %.3g) %s" % (err, name)
        for _ in range(10):
            A.hip_matvec_mult(bench.ALPHA, x.data_ptr(), y.data_ptr(), st_)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                A.hip_matvec_mult(bench.ALPHA, x.data_ptr(), y.data_ptr(), st_)
            e1.record()
            torch.cuda.synchronize()
            ts.append(1e-3 * e0.elapsed_time(e1) / args.steps)
        t = float(np.median(ts))
        xw = "xw %d KB LDS" % (info.unit_window_lds // 1024) if info.unit_windows else "-"
        if int(getattr(info, "sym_pipeline", 0)):
            xw = "sx %.1f %% of stored nnz" % (100.0 * info.sym_pipeline_elems / max(int(info.nnz_stored), 1))
        print("| %s | %s | %s | %.2f | %.1f | %.3f | %d | %d | %d, %s | %.2f | %.2f | %.3f |" % (
            label, "symmetric" if args.symmetric else "general", name if not body else "%s (`%s`)" % (name, body),
            1e6 * t, 2.0 * nnz / t / 1e9, info.index_bytes / max(int(info.nnz_stored), 1), int(info.nnz_stored),
            int(info.n_rowblocks), int(info.waves), xw, info.tune_seconds, info.emit_seconds, err), flush=True)
        A.destroy()


if __name__ == "__main__":
    main()
