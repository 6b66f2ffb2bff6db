"""Debug helper for a failing seed of tools/soak_large.py: which rows are wrong, under which options.
usage: python tools/soak_debug.py <seed> [key=value ...]"""
import os, sys
import numpy as np, scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.argv_saved = sys.argv[:]
seed = int(sys.argv[1]); extra = dict(a.split("=", 1) for a in sys.argv[2:])
sys.argv = [sys.argv[0], "0", "0"]
import importlib.util
spec = importlib.util.spec_from_file_location("soak_large", os.path.join(ROOT, "tools", "soak_large.py"))
sl = importlib.util.module_from_spec(spec)
try:
    spec.loader.exec_module(sl)
except SystemExit:
    pass
from sparsex_amd import synth
from helpers import tune
sym = seed % 3 != 0
csr, m = sl.big_matrix(seed, sym)
rp, ci, va, n = csr
o = sl.options(seed, sym)
o.update(extra)
print("seed", seed, "n", n, "nnz", m.nnz, "sym", sym, o)
A = tune(csr, o, sym=sym)
i = A.info()
print("rowblocks", i.n_rowblocks, "shared rows", i.n_shared_rows, "sym_tiles", i.sym_tiles, "segments", i.sym_segments, "waves", i.waves, "wave_tiles", i.wave_tiles)
x = synth.random_x(n)
for rep in range(3):
    y = np.full(n, np.nan)
    A.matvec_mult(0.5, x, y)
    want = 0.5 * (m @ x)
    err = np.abs(y - want)
    badrows = np.nonzero(~(err <= 1e-9 * np.maximum(np.abs(want), 1e-30) + 1e-12))[0]
    lens = np.diff(rp)
    print("rep", rep, "bad rows:", badrows.size, badrows[:12], "row lengths", lens[badrows[:12]], "err", err[badrows[:6]], "want", want[badrows[:6]], "nan", int(np.isnan(y).sum()))
y0 = synth.random_x(n, seed=seed + 1)
for rep in range(2):
    y = y0.copy()
    A.matvec_kernel(2.0, x, -0.5, y)
    want = 2.0 * (m @ x) - 0.5 * y0
    err = np.abs(y - want)
    badrows = np.nonzero(~(err <= 1e-9 * np.maximum(np.abs(want), 1e-30) + 1e-12))[0]
    lens = np.diff(rp)
    print("beta rep", rep, "bad rows:", badrows.size, badrows[:12], "row lengths", lens[badrows[:12]], "err", err[badrows[:6]], "want", want[badrows[:6]])
