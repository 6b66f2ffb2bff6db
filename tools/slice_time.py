"""Local SpMV time of single row slices of syn-nlpkkt as `world` ranks would hold them (load
balance of a multi-GPU run, measured on the one GPU of the test box): general path (rows dealt by
nonzeros) and symmetric path (rows dealt by stored nonzeros), every rank, plus the sizes of what
the step would exchange.
usage: tools/slice_time.py <world> [grid edge, default 120] [ranks, e.g. 0,3,7; default all]"""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import sparsex_amd as sx, bench
from sparsex_amd import synth
W = int(sys.argv[1]); N = int(sys.argv[2]) if len(sys.argv) > 2 else 120
ranks = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else list(range(W))
counts = synth.nlpkkt_row_counts(N); n = counts.size
stored = synth.stored_row_counts("nlpkkt", N, counts)
x = torch.from_numpy(synth.random_x(n)).cuda(); y = torch.zeros(n, dtype=torch.float64, device="cuda")
for sym in (False, True):
    cuts = bench.nnz_balanced_cuts(stored if sym else counts, W)
    for r in ranks:
        lo, hi = cuts[r], cuts[r + 1]
        rp, ci, va, _ = synth.syn_nlpkkt_rows(N, lo, hi, counts=counts)
        A = bench.tune((rp, ci, va, n), {"spx.rt.nr_threads": 16, "spx.rt.row_offset": lo, "spx.rt.global_rows": n,
                                          "spx.matrix.symmetric": "true" if sym else "false", "spx.rt.keep_encoded": "false"}, nrows=hi - lo)
        conflict = 0
        if sym:
            below = ci[ci < lo]
            conflict = int(np.unique(below).size)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(5): A.hip_matvec_mult(0.5, x.data_ptr(), y.data_ptr(), st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): A.hip_matvec_mult(0.5, x.data_ptr(), y.data_ptr(), st)
        e1.record(); torch.cuda.synchronize()
        i = A.info()
        print("edge %d world %d %s rank %d rows [%d, %d): %.1f us per local SpMV, %d row-blocks, %d nnz stored, segments %d; "
              "sends %d conflict rows (%.2f MB), receives %.1f MB of y slices in the hand-round" % (
                  N, W, "symmetric" if sym else "general", r, lo, hi, e0.elapsed_time(e1) * 1e3 / 50, i.n_rowblocks,
                  i.nnz_stored, i.sym_segments, conflict, 8e-6 * conflict, 8e-6 * (n - (hi - lo))), flush=True)
        A.destroy()
