#!/bin/bash
# quick ablations: prints us/step for several SPX_ABLATE masks (parity gate is skipped)
for w in syn-cant syn-nd24k; do for m in 0 1 2 3; do
SPX_ABLATE=$m python - $w $m <<'PY'
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bench, sparsex_amd as sx
from sparsex_amd import synth
w, m = sys.argv[1], sys.argv[2]
csr = bench.make_workload(w, 1.0)
A = bench.tune(csr, {"spx.rt.nr_threads": 8, "spx.gpu.rowblock_elems": 2048})
n = csr[3]
x = torch.from_numpy(synth.random_x(n)).cuda(); y = torch.zeros(n, dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(50): A.hip_matvec_mult(0.5, x.data_ptr(), y.data_ptr(), st)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(300): A.hip_matvec_mult(0.5, x.data_ptr(), y.data_ptr(), st)
torch.cuda.synchronize(); print(w, "ablate", m, "us/step %.2f" % ((time.perf_counter() - t0) / 300 * 1e6))
PY
done; done
