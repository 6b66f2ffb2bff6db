"""Bandwidth of the device-resident BLAS-1 helpers (HBM-bound streaming kernels)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsex_amd as sx
for n in [1 << 20, 1 << 24, 27993600, 1 << 27, 1 << 28]:
    a, b, c = sx.DeviceVector(n), sx.DeviceVector(n), sx.DeviceVector(n)
    a.init(1.0); b.init(2.0)
    def t(f, reps=50):
        for _ in range(5): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
    ts = {"scale_add (24 B/elem)": (t(lambda: a.scale_add_into(b, c, 0.5)), 24),
          "scale (16 B/elem)": (t(lambda: a.scale_into(c, 0.5)), 16),
          "copy (16 B/elem)": (t(lambda: a.copy_into(c)), 16),
          "dot (16 B/elem, incl. D2H of the result)": (t(lambda: a.dot(b)), 16),
          "dot with itself (8 B/elem, incl. D2H)": (t(lambda: a.dot(a)), 8)}
    for k, (sec, bpe) in ts.items():
        print("n=%9d %-42s %8.1f us  %7.1f GB/s" % (n, k, sec * 1e6, n * bpe / sec / 1e9))
