"""PCIe-inclusive rate of the drop-in entry point: spx_matvec_mult / spx_matvec_kernel
on HOST vectors (x up, [y up,] kernel, y down, synchronous), per workload."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sparsex_amd import synth

for w in ("syn-cant", "syn-nd24k", "syn-webbase"):
    csr = bench.make_workload(w, 1.0)
    n, nnz = csr[3], int(csr[0][-1])
    A = bench.tune(csr, {"spx.rt.nr_threads": 8, "spx.rt.keep_encoded": "false"})
    x, y = synth.random_x(n), np.zeros(n)
    for _ in range(20):
        A.matvec_mult(0.5, x, y)
    res = {}
    for name, fn in (("mult", lambda: A.matvec_mult(0.5, x, y)), ("kernel(beta=1)", lambda: A.matvec_kernel(0.5, x, 1.0, y))):
        t0 = time.perf_counter()
        for _ in range(200):
            fn()
        res[name] = (time.perf_counter() - t0) / 200
    print("%-12s n %8d  spx_matvec_mult %7.1f us (%6.1f GFLOP/s)   spx_matvec_kernel %7.1f us" %
          (w, n, res["mult"] * 1e6, 2.0 * nnz / res["mult"] / 1e9, res["kernel(beta=1)"] * 1e6))
