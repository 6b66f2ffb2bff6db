import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import sparsex_amd as sx, bench
from sparsex_amd import synth
# usage: tools/slice_time.py <world> [grid edge, default 120]
W = int(sys.argv[1]); N = int(sys.argv[2]) if len(sys.argv) > 2 else 120
counts = synth.nlpkkt_row_counts(N); n = counts.size
cuts = bench.nnz_balanced_cuts(counts, W)
x = torch.from_numpy(synth.random_x(n)).cuda(); y = torch.zeros(n, dtype=torch.float64, device="cuda")
for sym in (False, True):
    for r in (0, W // 2, W - 1):
        lo, hi = cuts[r], cuts[r + 1]
        rp, ci, va, _ = synth.syn_nlpkkt_rows(N, lo, hi, counts=counts)
        A = bench.tune((rp, ci, va, n), {"spx.rt.nr_threads": 16, "spx.rt.row_offset": lo, "spx.rt.global_rows": n,
                                          "spx.matrix.symmetric": "true" if sym else "false", "spx.rt.keep_encoded": "false"}, nrows=hi - lo)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(5): A.hip_matvec_mult(0.5, x.data_ptr(), y.data_ptr(), st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): A.hip_matvec_mult(0.5, x.data_ptr(), y.data_ptr(), st)
        e1.record(); torch.cuda.synchronize()
        i = A.info()
        print("edge %d world %d sym %d rank %d: %.1f us per local SpMV, %d row-blocks, %d nnz stored, segments %d" % (N, W, sym, r, e0.elapsed_time(e1) * 1e3 / 50, i.n_rowblocks, i.nnz_stored, i.sym_segments), flush=True)
        A.destroy()
