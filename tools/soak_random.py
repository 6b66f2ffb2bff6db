"""Extended run of the randomised parity tests (tests/test_gpu_random.py draws 40 + 50 seeds):
usage: python tools/soak_random.py <first seed> <last seed>   -- symmetric matrices, random options
incl. read-once segments / wide row-blocks / shortest run; product on the GPU against CSR."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sparsex_amd import synth
from helpers import tune, check_y
from test_stream_random import random_matrix, random_options, random_sym_options

a, b = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(a, b):
    sym = seed % 3 != 0
    csr, m = random_matrix(seed, symmetric=sym)
    n = csr[3]
    o = random_options(seed)
    if sym:
        o = random_sym_options(seed, o)
        if seed % 2:
            o["spx.gpu.sym_segments"] = "true"
    else:
        o["spx.gpu.rowblock_rows"] = str([512, 1024, 2048][seed % 3 if seed % 9 else 2])
    o["spx.gpu.waves"] = str([0, 2, 4, 8][seed % 4])
    # round 3: column slices (one launch / launched in turn), kept units, band launch order
    if not sym:
        o["spx.gpu.col_phases"] = ["1", "c2", "c4", "2", "3", "c8", "auto"][seed % 7]
    o["spx.gpu.keep_units"] = "false" if seed % 5 == 0 else "true"
    # round 5: the unit windows of x (forced on three times out of four; they apply where the stream is not sliced
    # and has no launch order), small and large budgets, gaps, joined row-blocks
    o["spx.gpu.unit_windows"] = ["true", "auto", "true", "false", "true"][seed % 5]
    o["spx.gpu.unit_window_doubles"] = str([64, 3072, 512, 8192, 1024, 12000][seed % 6])
    o["spx.gpu.unit_window_gap"] = str([0, 16, 200, 2][seed % 4])
    if not sym and seed % 7 == 3:
        o["spx.gpu.rowblock_elems"] = "16384"
    o["spx.gpu.band_order"] = "true" if seed % 4 == 1 else "false"
    try:
        A = tune(csr, o, sym=sym)
        x = synth.random_x(n)
        y = np.full(n, np.nan)
        A.matvec_mult(0.5, x, y)
        check_y(csr, x, y, 0.5)
        y0 = synth.random_x(n, seed=seed + 1)
        y = y0.copy()
        A.matvec_kernel(2.0, x, -0.5, y)
        check_y(csr, x, y, 2.0, -0.5, y0)
        A.destroy()
    except Exception as e:                     # keep going: report every failing seed
        bad += 1
        print("seed %d FAILED: %s %s" % (seed, type(e).__name__, str(e)[:200]), flush=True)
print("seeds [%d, %d): %d failures" % (a, b, bad))
sys.exit(1 if bad else 0)
