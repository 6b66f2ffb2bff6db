"""Randomised parity on larger matrices (20 k - 300 k rows: many row-blocks, slot windows that
overflow, wide row-blocks, x windows, 24-bit column offsets), on the GPU against CSR.
usage: python tools/soak_large.py <first seed> <last seed> [--host | --roundtrip | --library-vectors]   (--host: tune host-only
and check the decoded stream instead -- runs without a GPU; --roundtrip: also set entries, save, restore,
multiply again; SOAK_SIZES=1000000,3000000 for matrices big enough for the size-dependent choices,
e.g. read-once segments in "auto")"""
import os, sys
import numpy as np, scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sparsex_amd import synth
from helpers import tune, check_y

host = "--host" in sys.argv
a0, b0 = int(sys.argv[1]), int(sys.argv[2])


def big_matrix(seed, symmetric):
    rng = np.random.RandomState(7000 + seed)
    n = int(rng.choice([int(v) for v in os.environ.get("SOAK_SIZES", "20000,50000,120000,300000").split(",")]))
    rows, cols = [np.arange(n)], [np.arange(n)]
    for _ in range(rng.randint(2, 6)):
        kind = rng.randint(0, 6)
        if kind == 0:                                   # scatter
            k = rng.randint(n // 2, 3 * n)
            rows.append(rng.randint(0, n, k)); cols.append(rng.randint(0, n, k))
        elif kind == 1:                                 # runs of w columns at a few offsets (a stencil)
            w = rng.randint(2, 12)
            for off in rng.randint(-n // 3, n // 3, rng.randint(1, 5)):
                r = np.arange(max(0, -off), min(n, n - off - w))
                keep = rng.rand(r.size) > 0.02
                for q in range(w):
                    rows.append(r[keep]); cols.append(r[keep] + off + q)
        elif kind == 2:                                 # band of random nonzeros
            k = rng.randint(n, 6 * n)
            r = rng.randint(0, n, k)
            rows.append(r); cols.append(np.clip(r + rng.randint(-400, 401, k), 0, n - 1))
        elif kind == 3:                                 # aligned 8x8 tiles near a few block diagonals
            nb = n // 8
            for off in rng.randint(-nb // 4, nb // 4, rng.randint(1, 4)):
                i = np.arange(max(0, -off), min(nb, nb - off))
                i = i[rng.rand(i.size) < 0.3]
                a, b = np.meshgrid(np.arange(8), np.arange(8), indexing="ij")
                rows.append((8 * i[:, None] + a.ravel()[None, :]).ravel())
                cols.append((8 * (i + off)[:, None] + b.ravel()[None, :]).ravel())
        elif kind == 4:                                 # a few very long rows
            for _ in range(rng.randint(1, 3)):
                r = rng.randint(0, n)
                c = rng.choice(n, rng.randint(5000, 15000), replace=False)
                rows.append(np.full(c.size, r)); cols.append(c)
        else:                                           # 2x2 blocks shifting by two (the nlpkkt stand-in's shape)
            w = 2 * rng.randint(1, 4)
            off = 2 * rng.randint(-n // 8, n // 8)
            k = np.arange(max(0, -off // 2 + 1), min(n // 2, (n - off - w) // 2))
            for d in (0, 1):
                for q in range(w):
                    rows.append(2 * k + d); cols.append(2 * k + off + q)
    # round 6 (a generator of its own: the matrices of the seeds drawn so far gain these nonzeros, nothing else moves):
    # every third seed also holds a few CLEAN bands of two to four consecutive columns below the diagonal -- long
    # unbroken runs of equal segments, which fill read-once passes of their own (csx_spmv_sx_kernel)
    if seed % 3 == 1:
        r6 = np.random.RandomState(7600 + seed)
        for _ in range(r6.randint(1, 4)):
            w = r6.randint(2, 5)
            off = -int(r6.randint(w + 1, max(w + 2, n // 4)))
            lo = -off
            hi = n if r6.rand() < 0.5 else min(n, lo + int(r6.randint(100, 5000)))
            rr = np.arange(lo, hi)
            for q in range(w):
                rows.append(rr); cols.append(rr + off + q)
    r, c = np.concatenate(rows), np.concatenate(cols)
    ok = (r >= 0) & (r < n) & (c >= 0) & (c < n)
    r, c = r[ok], c[ok]
    if symmetric:
        r, c = np.concatenate([r, c]), np.concatenate([c, r])
    m = sp.coo_matrix((np.ones(r.size), (r, c)), shape=(n, n)).tocsr()
    m.sum_duplicates(); m.sort_indices()
    m.data = rng.uniform(0.5, 1.5, m.nnz)
    if symmetric:
        low = sp.tril(m, k=-1)
        m = (low + low.T + sp.diags(m.diagonal())).tocsr()
        m.sort_indices()
    return (m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.copy(), n), m


def options(seed, symmetric):
    rng = np.random.RandomState(9000 + seed)
    o = {"spx.rt.nr_threads": str(rng.choice([1, 3, 8]))}
    if rng.rand() < 0.5:
        o["spx.preproc.sampling"] = "none"
    if rng.rand() < 0.4:
        o["spx.gpu.rowblock_elems"] = str(rng.choice([700, 2000, 8192]))
    if rng.rand() < 0.3:
        o["spx.gpu.rowblock_rows"] = str(rng.choice([64, 512, 1024, 2048]))
    if symmetric:
        o["spx.gpu.sym_segments"] = str(rng.choice(["true", "true", "false", "auto"]))
        o["spx.gpu.sym_wide_rows"] = str(rng.choice([512, 1024, 2048]))
        o["spx.gpu.sym_segment_min"] = str(rng.choice([2, 3, 5]))
        if rng.rand() < 0.2:
            o["spx.gpu.sym_spill"] = str(rng.choice(["lists", "atomic"]))
        if rng.rand() < 0.1:
            o["spx.gpu.deterministic"] = "true"
    o["spx.gpu.waves"] = str(rng.choice([0, 2, 4, 8]))
    # round 3: column slices (one launch / launched in turn), kept units, band launch order
    if not symmetric and rng.rand() < 0.5:
        o["spx.gpu.col_phases"] = str(rng.choice(["c2", "c4", "c8", "2", "3", "auto"]))
    if rng.rand() < 0.2:
        o["spx.gpu.keep_units"] = "false"
    if rng.rand() < 0.3:
        o["spx.gpu.band_order"] = "true"
    # round 5: the unit windows (a generator of its own, so that the draws above stay what they were)
    r5 = np.random.RandomState(9500 + seed)
    o["spx.gpu.unit_windows"] = str(r5.choice(["true", "true", "auto", "false"]))
    o["spx.gpu.unit_window_doubles"] = str(r5.choice([256, 1024, 3072, 8192, 16384]))
    o["spx.gpu.unit_window_gap"] = str(r5.choice([0, 16, 100, 255]))
    # round 6: the read-once pipeline and the passes of their own (again a generator of its own)
    r6 = np.random.RandomState(9600 + seed)
    if symmetric:
        o["spx.gpu.sym_pipeline"] = str(r6.choice(["true", "true", "auto", "false"]))
        o["spx.gpu.sym_pure_passes"] = str(r6.choice(["true", "true", "false"]))
    return o


bad = 0
ran_sx = ran_xw = 0          # how many of the products ran through the pipelined kernels
by_need = 0                  # (--library-vectors) how many had x sent piece by piece in the order the parts needed it
sx_share = []
for seed in range(a0, b0):
    sym = seed % 3 != 0
    csr, m = big_matrix(seed, sym)
    n = csr[3]
    o = options(seed, sym)
    try:
        x = synth.random_x(n)
        if host:
            from stream_decode import Stream
            A = tune(csr, o, sym=sym, host_only=True)
            f = "/tmp/soak_large_%d.spx" % os.getpid()
            A.save(f)
            s = Stream(f)
            if not ((s.rbs["flags"] & 4) != 0).any():      # (column slices: several row-blocks per row by design)
                s.check_ownership()
            assert np.allclose(s.matvec(x), m @ x, rtol=1e-12, atol=1e-13), "decoded product"
        else:
            A = tune(csr, o, sym=sym)
            inf = A.info()
            ran_sx += int(getattr(inf, "sym_pipeline", 0) == 1)
            sx_share.append(inf.sym_pipeline_elems / max(int(inf.nnz_stored), 1) if getattr(inf, "sym_pipeline", 0) == 1 else 0.0)
            ran_xw += int(inf.unit_windows == 1)
            y = np.full(n, np.nan)
            A.matvec_mult(0.5, x, y)
            check_y(csr, x, y, 0.5)
            y0 = synth.random_x(n, seed=seed + 1)
            y = y0.copy()
            A.matvec_kernel(2.0, x, -0.5, y)
            check_y(csr, x, y, 2.0, -0.5, y0)
            if "--library-vectors" in sys.argv:
                # the same through vectors the library created (page-locked): with SPX_HOST_PARTS_MIN_BYTES and
                # SPX_HOST_XPIECE_BYTES lowered, x (and y, beta != 0) go up piece by piece in the order the parts need
                # them; a product with another x first, so that a piece the plan forgot shows
                import ctypes as C
                import sparsex_amd as sx
                from sparsex_amd.api import VectorStruct
                L = sx.lib()
                L.spx_vec_create_random.restype = C.POINTER(VectorStruct); L.spx_vec_create_random.argtypes = [C.c_size_t, C.c_void_p]
                L.spx_vec_create.restype = C.POINTER(VectorStruct); L.spx_vec_create.argtypes = [C.c_size_t, C.c_void_p]
                L.spx_mat_get_partition.restype = C.c_void_p
                L.spx_matvec_kernel.argtypes = [C.c_double, C.c_void_p, C.POINTER(VectorStruct), C.c_double, C.POINTER(VectorStruct)]
                L.spx_hip_mat_host_order.restype = C.c_int
                L.spx_hip_mat_host_order.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int]
                part = C.c_void_p(L.spx_mat_get_partition(C.c_void_p(A.handle)))
                xv, yv = L.spx_vec_create_random(n, part), L.spx_vec_create(n, part)
                xa = np.ctypeslib.as_array(xv.contents.elements, shape=(n,))
                ya = np.ctypeslib.as_array(yv.contents.elements, shape=(n,))
                xa[:] = np.nan
                assert L.spx_matvec_mult(C.c_double(0.5), C.c_void_p(A.handle), xv, yv) == 0
                xa[:] = x
                ya[:] = np.nan
                assert L.spx_matvec_mult(C.c_double(0.5), C.c_void_p(A.handle), xv, yv) == 0
                check_y(csr, x, ya.copy(), 0.5)
                buf = (C.c_int32 * 64)()
                by_need += int(L.spx_hip_mat_host_order(C.c_void_p(A.handle), buf, 64) >= 2)
                ya[:] = y0
                assert L.spx_matvec_kernel(2.0, C.c_void_p(A.handle), xv, -0.5, yv) == 0
                check_y(csr, x, ya.copy(), 2.0, -0.5, y0)
                L.spx_vec_destroy(xv); L.spx_vec_destroy(yv)
            if "--roundtrip" in sys.argv:
                # a few entries changed, saved, restored in place of the tuned matrix, multiplied again
                import sparsex_amd as sx
                rp, ci, va, _ = csr
                va2 = va.copy()
                rows = np.repeat(np.arange(n), np.diff(rp))
                rng = np.random.RandomState(seed)
                for j in rng.choice(rp[-1], size=20, replace=False):
                    r, c = int(rows[j]), int(ci[j])
                    A.set_entry(r, c, 2.5)
                    va2[j] = 2.5
                    if sym:
                        va2[rp[c] + int(np.searchsorted(ci[rp[c]:rp[c + 1]], r))] = 2.5
                f = "/tmp/soak_large_%d.spx" % os.getpid()
                A.save(f)
                A.destroy()
                sx.options_reset()
                A = sx.mat_restore(f)
                y = np.full(n, np.nan)
                A.matvec_mult(0.5, x, y)
                check_y((rp, ci, va2, n), x, y, 0.5)
                for j in rng.choice(rp[-1], size=20, replace=False):
                    assert A.get_entry(int(rows[j]), int(ci[j])) == va2[j]
        A.destroy()
        print("seed %d ok: n %d nnz %d sym %d %s" % (seed, n, m.nnz, sym, o), flush=True)
    except Exception as e:
        bad += 1
        print("seed %d FAILED: %s %s n %d %s" % (seed, type(e).__name__, str(e)[:200], n, o), flush=True)
print("seeds [%d, %d): %d failures; csx_spmv_sx_kernel ran on %d matrices (up to %.0f %% of their stored nonzeros in SX passes), "
      "csx_spmv_xw_kernel on %d; x by need on %d" % (a0, b0, bad, ran_sx, 100.0 * max(sx_share or [0.0]), ran_xw, by_need))
sys.exit(1 if bad else 0)
