#!/usr/bin/env python3
"""sha256 of the saved descriptor stream (spx_mat_save) of a fixed set of host-only tunes: a
change of the preprocessor or the emitter that is meant to leave the stream alone shows the same
hashes before and after.  usage: tools/stream_hash.py [--edge N] [--threads T]"""
import argparse, hashlib, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--edge", type=int, default=40)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--opt", action="append", default=[], help="extra option=value for every tune (ignored by a library that does not know it)")
    args = ap.parse_args()
    import sparsex_amd as sx
    from sparsex_amd import synth
    import bench
    cases = [("syn-nlpkkt e%d" % args.edge, synth._rows(bench.SLICED["syn-nlpkkt"], args.edge, 0, None, None,
                                                        synth.SEED_BASE + 4), (False, True)),
             ("syn-kkt2f e%d" % (args.edge // 2), synth._rows(bench.SLICED["syn-kkt2f"], args.edge // 2, 0, None, None,
                                                              synth.SEED_BASE + 4), (False, True)),
             ("syn-cant", synth.WORKLOADS["syn-cant"](0.25), (False, True)),
             ("syn-nd24k", synth.WORKLOADS["syn-nd24k"](0.1), (False, True)),
             ("syn-webbase", synth.WORKLOADS["syn-webbase"](0.25), (False,))]
    for name, csr, syms in cases:
        for sym in syms:
            for extra in ({}, {"spx.preproc.xform": "none"}):
                opts = {"spx.rt.nr_threads": args.threads, "spx.rt.host_only": "true",
                        "spx.matrix.symmetric": "true" if sym else "false"}
                opts.update(extra)
                for kv in args.opt:
                    k, v = kv.split("=", 1)
                    try:
                        sx.options_reset()
                        sx.option_set(k, v)
                        opts[k] = v
                    except Exception:
                        pass
                t0 = time.time()
                A = bench.tune(csr, opts)
                dt = time.time() - t0
                with tempfile.NamedTemporaryFile(suffix=".spx", dir="/tmp") as f:
                    A.save(f.name)
                    h = hashlib.sha256(open(f.name, "rb").read()).hexdigest()[:16]
                print("%-16s sym=%d %-28s %s  %.2f s" % (name, sym, extra or "", h, dt), flush=True)
                A.destroy()


if __name__ == "__main__":
    main()
