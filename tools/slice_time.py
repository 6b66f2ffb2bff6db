#!/usr/bin/env python3
"""One-GPU proxy of a multi-GPU step of syn-nlpkkt (the pool has no multi-GPU node): every slice
that `world` ranks would hold is generated, tuned and timed ON ITS OWN on the one GPU of the box --
the local product as one launch and in --parts launches (what SPX_DIST_OVERLAP interleaves its
rounds with) -- next to what the rank would exchange: its halo of x (general path; with --reorder
after spx_hip_dist_reorder), its conflict rows (symmetric path), the hand-round of whole slices.
The predicted step uses 50-75 GB/s per xGMI link and direction (MI355X: 7 links x ~153 GB/s
bidirectional per GPU, point to point): plain = product + largest per-peer message / link;
overlapped = max(product, messages) + the last round.

usage: tools/slice_time.py <world> [--edge N] [--ranks 0,3,7] [--reorder none|rcm|rcm_owner] [--parts K]
                           [--symmetric]   (markdown rows on stdout)
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("world", type=int)
    ap.add_argument("--edge", type=int, default=120)
    ap.add_argument("--ranks", default="")
    ap.add_argument("--reorder", default="none", choices=["none", "rcm", "rcm_owner"])
    ap.add_argument("--parts", type=int, default=4)
    ap.add_argument("--symmetric", action="store_true")
    ap.add_argument("--header", action="store_true")
    args = ap.parse_args()
    import torch
    import sparsex_amd as sx
    from sparsex_amd import synth
    import bench
    W, N = args.world, args.edge
    ranks = [int(v) for v in args.ranks.split(",")] if args.ranks else list(range(W))
    counts = synth.nlpkkt_row_counts(N)
    n = counts.size
    perm = inv = None
    counts_new = counts
    stored = synth.stored_row_counts("nlpkkt", N, counts) if args.symmetric else None
    if args.reorder != "none":
        rp_f, ci_f = synth._pattern("nlpkkt", N, counts)
        perm = sx.dist_reorder(rp_f, ci_f, n, W, sx.SPX_DIST_REORDER_RCM if args.reorder == "rcm" else sx.SPX_DIST_REORDER_RCM_OWNER,
                               pattern_symmetric=True)
        inv = np.empty(n, dtype=np.int64)
        inv[perm] = np.arange(n)
        counts_new = counts[inv]
        if args.symmetric:
            stored = np.zeros(n, dtype=np.int32)
            for r0 in range(0, n, 1 << 20):
                r1 = min(n, r0 + (1 << 20))
                rows = np.repeat(np.arange(r0, r1, dtype=np.int64), counts[r0:r1])
                below = perm[ci_f[rp_f[r0]:rp_f[r1]]] <= perm[rows]
                stored[perm[r0:r1]] = np.bincount((rows - r0)[below], minlength=r1 - r0)
        del rp_f, ci_f
    cuts = bench.nnz_balanced_cuts(stored if args.symmetric else counts_new, W)
    x = torch.from_numpy(synth.random_x(n)).cuda()
    y = torch.zeros(n, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    if args.header:
        print("| edge | ranks | numbering | path | rank | rows | local product us (one launch) | in %d launches us | halo of x received "
              "(entries, MB) | largest message with one peer MB | whole-slice hand-round MB | predicted step us, plain "
              "(50 / 75 GB/s links) | overlapped |" % args.parts)
        print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in ranks:
        lo, hi = cuts[r], cuts[r + 1]
        if perm is None:
            rp, ci, va, _ = synth.syn_nlpkkt_rows(N, lo, hi, counts=counts)
        else:
            rp, ci, va, _ = synth._rows_perm("nlpkkt", N, inv[lo:hi], perm, counts, synth.SEED_BASE + 4)
        A = bench.tune((rp, ci, va, n), {"spx.rt.nr_threads": 16, "spx.rt.row_offset": lo, "spx.rt.global_rows": n,
                                          "spx.matrix.symmetric": "true" if args.symmetric else "false",
                                          "spx.rt.keep_encoded": "false"}, nrows=hi - lo)
        if args.symmetric:
            rows = np.repeat(np.arange(lo, hi, dtype=np.int64), np.diff(rp))
            need = np.unique(ci[(ci < lo) & (ci <= rows)])
        else:
            need = np.unique(ci[(ci < lo) | (ci >= hi)])
        owners = np.searchsorted(np.asarray(cuts[1:]), need, side="right")
        big = int(np.bincount(owners, minlength=W).max()) if need.size else 0

        def timed(fn, reps=50):
            for _ in range(5):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / reps
        t1 = timed(lambda: A.hip_matvec_mult(0.5, x.data_ptr(), y.data_ptr(), st))
        launched = [1]

        def parts():
            launched[0] = A.hip_matvec_parts(0.5, x.data_ptr(), 0.0, y.data_ptr(), args.parts, st)
        tk = timed(parts) if not args.symmetric else float("nan")
        msg = [8.0 * big / bw * 1e-3 for bw in (50.0, 75.0)]          # us at 50 / 75 GB/s (bytes / (GB/s) = ns)
        plain = [t1 + m for m in msg]
        over = [max(tk, m) + m / max(launched[0], 1) for m in msg] if not args.symmetric else plain
        print("| %d | %d | %s | %s | %d | %d | %.1f | %s | %d (%.2f) | %.2f | %.1f | %.0f / %.0f | %s |" % (
            N, W, args.reorder, "symmetric" if args.symmetric else "general", r, hi - lo, t1,
            "%.1f (%d)" % (tk, launched[0]) if not args.symmetric else "-", need.size, 8e-6 * need.size, 8e-6 * big,
            8e-6 * (n - (hi - lo)), plain[0], plain[1],
            "%.0f / %.0f" % (over[0], over[1]) if not args.symmetric else "-"), flush=True)
        A.destroy()


if __name__ == "__main__":
    main()
