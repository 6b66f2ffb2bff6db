"""Interleaved A/B timing of several option sets on one workload (one process).
usage: python tools/sweep.py <workload> key=v1,v2,... [key2=...]   (cartesian product)
Prints the median and min us/step of ROUNDS rounds of STEPS steps per config."""
import itertools, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sparsex_amd import synth

wl = sys.argv[1]
axes = []
for arg in sys.argv[2:]:
    k, vs = arg.split("=")
    axes.append([(k, v) for v in vs.split(",")])
ROUNDS = int(os.environ.get("ROUNDS", 5)); STEPS = int(os.environ.get("STEPS", 200))
csr = bench.make_workload(wl, 1.0)
n = csr[3]
x = torch.from_numpy(synth.random_x(n)).cuda()
y = torch.zeros(n, dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
cfgs = []
for combo in itertools.product(*axes) if axes else [()]:
    opts = {"spx.rt.nr_threads": 8, "spx.rt.keep_encoded": "false"}
    opts.update(dict(combo))
    A = bench.tune(csr, opts)
    cfgs.append((dict(combo), A, []))
for _, A, _ in cfgs:
    for _ in range(30): A.hip_matvec_mult(0.5, x.data_ptr(), y.data_ptr(), st)
torch.cuda.synchronize()
for r in range(ROUNDS):
    for combo, A, ts in cfgs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(STEPS): A.hip_matvec_mult(0.5, x.data_ptr(), y.data_ptr(), st)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / STEPS)
for combo, A, ts in cfgs:
    inf = A.info()
    print("%-14s %-50s median %7.2f us  min %7.2f  rb %6d idxB/nnz %.3f" % (
        wl, combo, float(np.median(ts)), min(ts), inf.n_rowblocks, inf.index_bytes / max(inf.nnz_stored, 1)))
