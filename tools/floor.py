import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sparsex_amd import synth
for scale in [0.001, 0.01, 0.05, 0.1, 0.25, 0.5, 1.0]:
    csr = bench.make_workload("syn-cant", scale)
    A = bench.tune(csr, {"spx.rt.nr_threads": 8, "spx.rt.keep_encoded": "false"})
    n = csr[3]
    x = torch.from_numpy(synth.random_x(n)).cuda(); y = torch.zeros(n, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(50): A.hip_matvec_mult(0.5, x.data_ptr(), y.data_ptr(), st)
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300): A.hip_matvec_mult(0.5, x.data_ptr(), y.data_ptr(), st)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / 300)
    inf = A.info()
    print("scale %.3f nnz %8d rb %5d  %.2f us/step" % (scale, inf.nnz_stored, inf.n_rowblocks, np.median(ts)))
