"""Regenerates tests/golden/reference_matrices.json from the reference's test
fixtures (test/matrices/*.mtx.sorted in the reference tree): the matrices the
reference's own test-suite runs, as zero-based CSR (symmetric files mirrored
the way include/sparsex/internals/Mmf.hpp:445-478 does).  Data only."""
import json
import os
import sys

import scipy.sparse as sp

REF = os.environ.get("SPX_REFERENCE_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def load(path):
    with open(path) as f:
        first = f.readline()
    sym = first.startswith("%%") and "symmetric" in first
    lines = [l for l in open(path) if l.strip() and not l.startswith("%")]
    n, m, _ = [int(float(t)) for t in lines[0].split()]
    r, c, v = [], [], []
    for l in lines[1:]:
        a, b, val = l.split()
        r.append(int(a) - 1); c.append(int(b) - 1); v.append(float(val))
        if sym and a != b:
            r.append(int(b) - 1); c.append(int(a) - 1); v.append(float(val))
    A = sp.csr_matrix((v, (r, c)), shape=(n, m))
    A.sort_indices()
    return {"n": n, "rowptr": A.indptr.tolist(), "colind": A.indices.tolist(),
            "values": A.data.tolist()}


if __name__ == "__main__":
    out = {}
    for name in ["demopatt", "symmetric", "symmetric-very-sparse", "test", "test2", "test3"]:
        out[name] = load(os.path.join(REF, "test", "matrices", name + ".mtx.sorted"))
    with open(os.path.join(HERE, "reference_matrices.json"), "w") as f:
        json.dump(out, f)
    print({k: (v["n"], len(v["values"])) for k, v in out.items()})
