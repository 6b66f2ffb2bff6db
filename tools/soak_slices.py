"""Randomised check of row slices on the host (no GPU): random matrices cut into 2-4 slices, every
slice tuned on its own (spx.rt.row_offset), its saved stream decoded by tests/stream_decode.py;
the slices' partial vectors must sum to A x and their conflict rows (spx_hip_mat_dist_plan needs a
transport, so: the rows in front of the slice that the decoded stream adds to) must be the
columns in front of the slice that its lower triangle touches.
usage: python tools/soak_slices.py <first seed> <last seed>"""
import os, sys
import numpy as np, scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sparsex_amd as sx
from sparsex_amd import synth
from stream_decode import Stream
from test_stream_random import random_matrix, random_options, random_sym_options
from test_row_slices import nnz_balanced_bounds

a0, b0 = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(a0, b0):
    sym = seed % 4 != 0
    csr, m = random_matrix(seed, symmetric=sym)
    rp, ci, va, n = csr
    world = 2 + seed % 3
    cuts = nnz_balanced_bounds(np.diff(rp), world)
    o = random_options(seed)
    if sym:
        o = random_sym_options(seed, o)
        if seed % 2:
            o["spx.gpu.sym_segments"] = "true"
    x = synth.random_x(n)
    y = np.zeros(n)
    try:
        for r in range(world):
            lo, hi = cuts[r], cuts[r + 1]
            if hi == lo:
                continue
            rl = (rp[lo:hi + 1] - rp[lo]).astype(np.int32)
            cl, vl = ci[rp[lo]:rp[hi]].copy(), va[rp[lo]:rp[hi]].copy()
            sx.options_reset()
            oo = dict(o)
            oo.update({"spx.rt.row_offset": lo, "spx.rt.global_rows": n, "spx.rt.host_only": "true",
                       "spx.matrix.symmetric": "true" if sym else "false"})
            for k, v in oo.items():
                sx.option_set(k, str(v))
            A = sx.mat_tune(sx.input_load_csr(rl, cl, vl, hi - lo, n))
            f = "/tmp/soak_slice_%d.spx" % os.getpid()
            A.save(f)
            s = Stream(f)
            part = s.matvec(x)
            assert not np.any(part[hi:]), "a slice adds below its rows"
            if not sym:
                assert not np.any(part[:lo])
            y += part
            sx.mat_restore(f).destroy()
            A.destroy()
        assert np.allclose(y, m @ x, rtol=1e-12, atol=1e-13), "slices do not sum to the product"
    except Exception as e:
        bad += 1
        print("seed %d FAILED: %s %s %s" % (seed, type(e).__name__, str(e)[:150], o), flush=True)
print("seeds [%d, %d): %d failures" % (a0, b0, bad))
sys.exit(1 if bad else 0)
