#!/usr/bin/env python3
"""What travels between the processes of a row-partitioned matrix, per rank, with and without the
partition-aware numbering (spx_hip_dist_reorder): host only, from the pattern.

For every (world, numbering): the rows and nonzeros of every rank (cut by nonzeros,
SparseInternal.hpp:131-144), its halo of x on the general path (distinct columns outside its rows that
its rows read = what SPX_DIST_HALO_X brings it; the gloo tests assert that the library's lists equal
exactly this set), and on the symmetric path (cut by stored nonzeros) its conflict rows (distinct
columns in front of its rows in its lower triangle = what it sends to the owners, CsxBuild.hpp:400-581).
Output: a markdown table.

usage: tools/dist_volumes.py [--workload syn-nlpkkt] [--edge 120] [--worlds 2,4,8] [--modes none,rcm,rcm_owner]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def permute_pattern(rp, ci, perm):
    """Pattern of P A P^T (rows in new order; columns renumbered, not sorted: only sets matter here)."""
    n = rp.size - 1
    inv = np.empty(n, dtype=np.int64)
    inv[perm] = np.arange(n)
    cnt = np.diff(rp)[inv]
    rp2 = np.concatenate([[0], np.cumsum(cnt, dtype=np.int64)])
    ci2 = np.empty(ci.size, dtype=np.int32)
    step = 1 << 20
    for a in range(0, n, step):
        b = min(n, a + step)
        old = inv[a:b]
        idx = np.concatenate([np.arange(rp[o], rp[o + 1]) for o in old]) if b - a < 64 else None
        if idx is None:
            # gather the old rows' column ranges without a python loop per row
            starts = rp[old].astype(np.int64)
            lens = cnt[a:b].astype(np.int64)
            offs = np.repeat(starts - np.concatenate([[0], np.cumsum(lens)[:-1]]), lens) + np.arange(int(lens.sum()))
            idx = offs
        ci2[rp2[a]:rp2[b]] = perm[ci[idx]]
    return rp2, ci2


def volumes(rp, ci, world, symmetric):
    import bench
    n = rp.size - 1
    counts = np.diff(rp)
    stored = bench.stored_counts_csr(rp, ci) if symmetric else None
    cuts = bench.nnz_balanced_cuts(stored if symmetric else counts, world)
    out = []
    for g in range(world):
        lo, hi = cuts[g], cuts[g + 1]
        cols = ci[rp[lo]:rp[hi]]
        if symmetric:
            rows = np.repeat(np.arange(lo, hi, dtype=np.int64), counts[lo:hi])
            cols = cols[cols <= rows]
            need = np.unique(cols[cols < lo])
        else:
            need = np.unique(cols[(cols < lo) | (cols >= hi)])
        owners = np.searchsorted(np.asarray(cuts[1:]), need, side="right")
        out.append({"rows": hi - lo, "nnz": int((stored if symmetric else counts)[lo:hi].sum()), "entries": int(need.size),
                    "peers": int(np.unique(owners).size),
                    "largest_peer": int(np.bincount(owners, minlength=world).max()) if need.size else 0})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="syn-nlpkkt")
    ap.add_argument("--edge", type=int, default=120)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--worlds", default="2,4,8")
    ap.add_argument("--modes", default="none,rcm,rcm_owner")
    args = ap.parse_args()
    import sparsex_amd as sx
    from sparsex_amd import synth
    import bench
    if args.workload in bench.SLICED:
        rp, ci, va, n = synth._rows(bench.SLICED[args.workload], args.edge, 0, None, None, synth.SEED_BASE + 4)
        label = "%s e%d" % (args.workload, args.edge)
    else:
        rp, ci, va, n = synth.WORKLOADS[args.workload](args.scale)
        label = args.workload
    del va
    rp = rp.astype(np.int64)
    print("`%s`: %d rows, %d nonzeros.  Entries are doubles (8 bytes); `largest peer` = most entries exchanged with ONE "
          "other rank (one xGMI link).\n" % (label, n, rp[-1]))
    print("| ranks | numbering | path | rows per rank (min / max) | halo of x received per rank (min / max) | "
          "largest peer | peers | whole-slice hand-round per rank | seconds to compute the numbering |")
    print("|---|---|---|---|---|---|---|---|---|")
    for world in [int(w) for w in args.worlds.split(",")]:
        for mode in args.modes.split(","):
            t0 = time.time()
            if mode == "none":
                rp2, ci2 = rp, ci
            else:
                perm = sx.dist_reorder(rp, ci, n, world, sx.SPX_DIST_REORDER_RCM if mode == "rcm" else sx.SPX_DIST_REORDER_RCM_OWNER,
                                       pattern_symmetric=args.workload != "syn-webbase")
                assert np.array_equal(np.sort(perm), np.arange(n))
                rp2, ci2 = permute_pattern(rp, ci, perm.astype(np.int64))
            dt = time.time() - t0
            for symmetric in ([False, True] if args.workload in bench.SYMMETRIC_WORKLOADS else [False]):
                v = volumes(rp2, ci2, world, symmetric)
                rows = [r["rows"] for r in v]
                ent = [r["entries"] for r in v]
                print("| %d | %s | %s | %d / %d | %d / %d (%.2f / %.2f MB) | %d | %d | %.2f MB | %.1f |" % (
                    world, mode, "symmetric: conflict rows sent" if symmetric else "general: halo received",
                    min(rows), max(rows), min(ent), max(ent), 8e-6 * min(ent), 8e-6 * max(ent),
                    max(r["largest_peer"] for r in v), max(r["peers"] for r in v),
                    8e-6 * (n - min(rows)), dt), flush=True)


if __name__ == "__main__":
    main()
