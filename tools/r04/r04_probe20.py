import os, sys, time
sys.path.insert(0, ".")
import torch
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
import sparsex_amd as sx
from sparsex_amd import synth
import bench
sx.lib().spx_log_info_console()
csr = synth.syn_nlpkkt_rows(40)
n = csr[3]
x = torch.from_numpy(synth.random_x(n)).cuda()
y = torch.zeros(n, dtype=torch.float64, device="cuda")
for rep in range(2):
    t = time.time()
    A = bench.tune(csr, {"spx.gpu.rowblock_elems": "1024", "spx.rt.nr_threads": "4"})
    t1 = time.time()
    k = A.hip_matvec_parts(0.5, x.data_ptr(), -0.75, y.data_ptr(), 2, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t2 = time.time()
    S = bench.tune(csr, {"spx.rt.nr_threads": "4", "spx.matrix.symmetric": "true"})
    t3 = time.time()
    print("== huge pages %s rep %d: tune %.2f s, product in parts %.2f s, symmetric tune %.2f s" % (
        "off" if os.environ.get("SPX_NO_HUGE_PAGES") else "on", rep, t1 - t, t2 - t1, t3 - t2), flush=True)
    A.destroy(); S.destroy()
