"""Randomised parity on rectangular matrices (rows != columns) and on matrices with empty rows /
columns at either end, general path, on the GPU against CSR.
usage: python tools/soak_rect.py <first seed> <last seed>"""
import os, sys
import numpy as np, scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sparsex_amd as sx
from sparsex_amd import synth
from helpers import FP64_BOUND_FACTOR
from test_stream_random import random_matrix, random_options

a0, b0 = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(a0, b0):
    rng = np.random.RandomState(50000 + seed)
    _, m = random_matrix(seed, symmetric=False)
    n = m.shape[0]
    nr = int(rng.randint(max(2, n // 3), n + 1)); nc = int(rng.randint(max(2, n // 3), n + 1))
    m = m[:nr, :nc].tocsr()
    if rng.rand() < 0.3:                                 # a block of empty rows at the end / start
        k = int(rng.randint(1, max(2, nr // 4)))
        z = sp.csr_matrix((k, nc))
        m = sp.vstack([z, m[k:]] if rng.rand() < 0.5 else [m[:-k], z]).tocsr()
    m.sort_indices()
    if m.nnz == 0:
        continue
    o = random_options(seed)
    o["spx.gpu.rowblock_rows"] = str([3, 512, 1024, 2048][seed % 4])
    o["spx.gpu.col_phases"] = ["1", "c2", "c4", "3", "c8", "auto"][seed % 6]        # round 3
    try:
        sx.options_reset()
        for k, v in o.items():
            sx.option_set(k, str(v))
        A = sx.mat_tune(sx.input_load_csr(m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.copy(), nr, nc))
        x = synth.random_x(nc)
        y0 = synth.random_x(nr, seed=seed + 1)
        y = y0.copy()
        A.matvec_kernel(2.0, x, -0.5, y)
        want = 2.0 * (m @ x) - 0.5 * y0
        bound = FP64_BOUND_FACTOR * 2.0 ** -53 * (2.0 * (abs(m) @ np.abs(x)) + 0.5 * np.abs(y0)) + 1e-300
        assert np.all(np.abs(y - want) <= bound), "kernel beta path"
        y = np.full(nr, np.nan)
        A.matvec_mult(0.5, x, y)
        assert np.all(np.abs(y - 0.5 * (m @ x)) <= bound), "mult"
        A.destroy()
    except Exception as e:
        bad += 1
        print("seed %d FAILED: %s %s shape %s %s" % (seed, type(e).__name__, str(e)[:200], m.shape, o), flush=True)
print("seeds [%d, %d): %d failures" % (a0, b0, bad))
sys.exit(1 if bad else 0)
