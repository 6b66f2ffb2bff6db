#!/usr/bin/env python3
"""Run-to-run spread of the bench matrix's product.

  spread_probe.py save <file> [--edge N]     tune syn-nlpkkt once, save the tuned stream
  spread_probe.py time <file> [--steps K]    restore it in THIS process and time the product
  spread_probe.py soak <file> --seconds S    ... batch after batch for S seconds ("SOAK <unix time> <us>")

`time` prints one JSON line: the per-batch times (HIP events on the launch stream), where the
arrays landed (virtual addresses), the read roof of the same process.
It is meant to be started several times in a row, plain and under `rocprofv3 --pmc ...
--kernel-trace` (the kernel's duration in the trace tells which group the process fell into,
the counters of the same dispatches what differs)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["save", "time", "soak", "place"])
    ap.add_argument("file")
    ap.add_argument("--edge", type=int, default=240)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--batches", type=int, default=5)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--symmetric", action="store_true")
    ap.add_argument("--tag", default="")
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--gap", type=float, default=0.0, help="soak: idle seconds every --gap-every batches")
    ap.add_argument("--gap-every", type=int, default=50)
    args = ap.parse_args()

    import torch
    import sparsex_amd as sx
    from sparsex_amd import synth
    import bench

    torch.cuda.set_device(0)
    if args.mode == "save":
        csr = synth._rows("nlpkkt", args.edge, 0, None, None, synth.SEED_BASE + 4)
        A = bench.tune(csr, {"spx.rt.nr_threads": args.threads, "spx.rt.keep_encoded": "false",
                             "spx.matrix.symmetric": "true" if args.symmetric else "false"})
        info = A.info()
        A.save(args.file)
        print(json.dumps({"saved": args.file, "nnz": int(info.nnz), "waves": int(info.waves),
                          "tune_s": info.tune_seconds, "emit_s": info.emit_seconds}), flush=True)
        return

    if args.mode == "place":
        # the same stream uploaded again and again inside ONE process, with allocations of other sizes
        # kept in between, so that every upload lands on other physical pages: does the level move?
        keep = []
        n = None
        for i, hold_mb in enumerate([0, 64, 1024, 3, 4096, 200, 16384, 1]):
            if hold_mb:
                keep.append(torch.empty(hold_mb << 17, dtype=torch.float64, device="cuda"))
            A = sx.mat_restore(args.file)
            if n is None:
                n = A.nrows
                x = torch.from_numpy(synth.random_x(n)).cuda()
                y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(5):
                A.hip_matvec_mult(bench.ALPHA, x.data_ptr(), y.data_ptr(), st)
            ts = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.steps):
                    A.hip_matvec_mult(bench.ALPHA, x.data_ptr(), y.data_ptr(), st)
                e1.record()
                torch.cuda.synchronize()
                ts.append(round(1e3 * e0.elapsed_time(e1) / args.steps, 2))
            print("PLACE upload %d after holding %d MB more: %s us" % (i, hold_mb, ts), flush=True)
            A.destroy()
        return

    t0 = time.perf_counter()
    A = sx.mat_restore(args.file)
    t_restore = time.perf_counter() - t0
    info = A.info()
    n = A.nrows
    xh = synth.random_x(n)
    x = torch.from_numpy(xh).cuda()
    y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        A.hip_matvec_mult(bench.ALPHA, x.data_ptr(), y.data_ptr(), st)
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.batches):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            A.hip_matvec_mult(bench.ALPHA, x.data_ptr(), y.data_ptr(), st)
        e1.record()
        torch.cuda.synchronize()
        ts.append(round(1e3 * e0.elapsed_time(e1) / args.steps, 2))
    if args.mode == "soak":
        # a time series inside ONE process: batch after batch for --seconds, with idle gaps
        # (the clocks are sampled by the calling shell: tools/r04/r04_probe2.sh)
        t_end = time.time() + args.seconds
        k = 0
        while time.time() < t_end:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                A.hip_matvec_mult(bench.ALPHA, x.data_ptr(), y.data_ptr(), st)
            e1.record()
            torch.cuda.synchronize()
            print("SOAK %.3f %.2f" % (time.time(), 1e3 * e0.elapsed_time(e1) / args.steps), flush=True)
            k += 1
            if args.gap and k % args.gap_every == 0:
                time.sleep(args.gap)
    peak = bench.measured_read_peak(sx, torch, elems=1 << 27, reps=6)
    ysum = float(torch.nan_to_num(y).abs().sum())
    print(json.dumps({"tag": args.tag, "us_per_spmv": ts, "median_us": float(np.median(ts)),
                      "read_peak_GBs": round(peak, 1), "restore_s": round(t_restore, 1),
                      "waves": int(info.waves), "x_ptr": hex(x.data_ptr()), "y_ptr": hex(y.data_ptr()),
                      "arena": os.environ.get("SPX_NO_ARENA") is None, "ysum": ysum,
                      }), flush=True)
    A.destroy()


if __name__ == "__main__":
    main()
