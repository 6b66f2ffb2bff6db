#!/usr/bin/env python3
"""Dumps a workload as a STANDARD Matrix Market file, the way SuiteSparse ships its matrices:
`%%MatrixMarket matrix coordinate real symmetric|general`, one-based, column-ordered; a symmetric
file holds the lower triangle (row >= column) only.  What spx_input_load_mmf() must cope with at
size (reference reader: include/sparsex/internals/Mmf.hpp:331-478 -- mirror the stored triangle,
sort row-major).

usage: tools/mm_write.py <workload> <out.mtx> [--edge N] [--scale S] [--general]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_mtx(path, csr, symmetric):
    """csr = (rowptr, colind, values, n) zero-based; symmetric: the matrix equals its transpose
    and only its lower triangle is written.  Returns the number of entries written."""
    from sparsex_amd import synth
    import scipy.sparse as sp
    rp, ci, va, n = csr
    a = sp.csr_matrix((va, ci, rp), shape=(n, n))
    if symmetric:
        # column-ordered lower triangle = row-ordered upper triangle with the roles swapped
        u = sp.triu(a, format="csr")
        u.sort_indices()
        cols = np.repeat(np.arange(n, dtype=np.int32), np.diff(u.indptr))      # the column of the file
        rows = u.indices.astype(np.int32)                                     # its row (>= column)
        vals = u.data
    else:
        c = a.tocsc()
        c.sort_indices()
        cols = np.repeat(np.arange(n, dtype=np.int32), np.diff(c.indptr))
        rows = c.indices.astype(np.int32)
        vals = c.data
    rows, cols, vals = np.ascontiguousarray(rows), np.ascontiguousarray(cols), np.ascontiguousarray(vals, dtype=np.float64)
    L = synth._synlib()
    L.spx_mm_write.restype = C.c_int64
    L.spx_mm_write.argtypes = [C.c_char_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    got = L.spx_mm_write(str(path).encode(), 1 if symmetric else 0, n, n, rows.size, rows.ctypes.data, cols.ctypes.data,
                         vals.ctypes.data)
    assert got == rows.size, "writing %s failed" % path
    return int(got)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("out")
    ap.add_argument("--edge", type=int, default=120)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--general", action="store_true", help="write every entry (banner 'general') even if the matrix is symmetric")
    args = ap.parse_args()
    from sparsex_amd import synth
    import bench
    if args.workload in bench.SLICED:
        csr = synth._rows(bench.SLICED[args.workload], args.edge, 0, None, None, synth.SEED_BASE + 4)
    else:
        csr = synth.WORKLOADS[args.workload](args.scale)
    sym = args.workload in bench.SYMMETRIC_WORKLOADS and not args.general
    k = write_mtx(args.out, csr, sym)
    print("%s: %d x %d, %d entries written (%s), %.1f MB" % (args.out, csr[3], csr[3], k, "symmetric, lower triangle" if sym else "general",
                                                           os.path.getsize(args.out) / 1e6))


if __name__ == "__main__":
    main()
