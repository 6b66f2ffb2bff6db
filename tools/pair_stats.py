#!/usr/bin/env python3
"""How many unit passes of a tuned stream share their x with their neighbour (host only).
usage: tools/pair_stats.py <workload> [--edge N] [--scale S] [--opt k=v ...]"""
import argparse, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload"); ap.add_argument("--edge", type=int, default=60); ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--opt", action="append", default=[])
    args = ap.parse_args()
    import sparsex_amd as sx
    from sparsex_amd import synth
    import bench
    from stream_decode import Stream
    if args.workload in bench.SLICED:
        csr = synth._rows(bench.SLICED[args.workload], args.edge, 0, None, None, synth.SEED_BASE + 4)
    else:
        csr = synth.WORKLOADS[args.workload](args.scale)
    opts = {"spx.rt.host_only": "true", "spx.rt.nr_threads": 8, "spx.rt.keep_encoded": "false"}
    for o in args.opt:
        k, v = o.split("=", 1); opts[k] = v
    A = bench.tune(csr, opts)
    f = tempfile.mktemp(suffix=".spx"); A.save(f); s = Stream(f); os.unlink(f)
    tot = shared = unit = lanes = inline = 0
    for rb in s.rbs:
        ps = s.passes[int(rb["pass_off"]):int(rb["pass_off"]) + int(rb["n_pass"])]
        tot += len(ps)
        for p in ps:
            if p["kind"] == 0:
                unit += 1; lanes += int(p["nseg"]); inline += int(p["flags"]) & 1
        for i in range(0, len(ps) - 1, 2):
            a, b = ps[i], ps[i + 1]
            if a["kind"] == 0 and b["kind"] == 0 and (int(a["flags"]) & int(b["flags"]) & 1) and a["nseg"] == b["nseg"] and a["width"] == b["width"]:
                ma, mb = int(a["mask"]), int(b["mask"])
                ba, bb = ma >> 32, mb >> 32
                if (ma & 0xffffffff) == (mb & 0xffffffff) and (ba >> 22) == (bb >> 22) and \
                   ((int(a["seg0"]) - ((ba >> 9) & 8191)) & 0xffff) == ((int(b["seg0"]) - ((bb >> 9) & 8191)) & 0xffff):
                    shared += 2
    info = A.info()
    print("%s: row-blocks %d passes %d unit passes %d (inline %d, lanes used %.1f %%) in couples that share x: %d (%.1f %% of the unit passes); index %.3f B/nnz" % (
        args.workload, len(s.rbs), tot, unit, inline, 100.0 * lanes / max(64 * unit, 1), shared, 100.0 * shared / max(unit, 1),
        info.index_bytes / max(int(info.nnz_stored), 1)))

if __name__ == "__main__":
    main()
