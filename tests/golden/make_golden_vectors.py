"""Generates tests/golden/csx_vectors.json: (ctl, values, id_map, x, alpha) -> y
known-answer vectors produced by the REFERENCE'S OWN SpMV templates, compiled
in place from the reference tree by oracle/build_ref.py (run in the authoring
container only; the reference is not available on the GPU box).

Two families:
  * hand-built streams, one per unit type (delta8/16/32, horizontal, vertical,
    diagonal, anti-diagonal, block-row RxC, block-col RxC, row jumps,
    full_colind, a non-zero row_start), including the worked example of
    SURVEY.md section 7.0;
  * streams emitted by this repository's preprocessor for the reference's test
    matrices under the reference's own test scenarios
    (test/scripts/test-sparsex.sh.in:55-244), general and symmetric.
Every vector stores the stream and the reference's y; the oracle
(oracle/csx_oracle.c) must reproduce y bit for bit (tests/test_oracle_golden.py).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import sparsex_amd as sx          # noqa: E402
from oracle import pyoracle       # noqa: E402


def varint(v):
    out = []
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        if v < 0x80:
            out.append(b)
            break
        out.append(b | 0x80)
        v >>= 7
    return out


def unit(slot, size, ucol, nr=False, rjmp=0, deltas=(), dbytes=1, full=False):
    b = [slot | (0x80 if nr else 0) | (0x40 if rjmp else 0), size]
    if rjmp:
        b += varint(rjmp)
    b += list(int(ucol).to_bytes(4, "little")) if full else varint(ucol)
    for d in deltas:
        b += list(int(d).to_bytes(dbytes, "little"))
    return b


def export(ctl, values, id_map, nrows, ncols, row_start=0, row_jumps=0, full_colind=0,
           dvalues=None):
    ids = list(id_map) + [-1] * (64 - len(id_map))
    return {"values": np.array(values, dtype=np.float64), "ctl": np.array(ctl, dtype=np.uint8),
            "nnz": len(values), "ncols": ncols, "nrows": nrows, "row_start": row_start,
            "row_jumps": row_jumps, "full_colind": full_colind, "id_map": ids,
            "rows_info": None, "dvalues": None if dvalues is None else np.array(dvalues)}


def hand_built():
    cases = []
    x12 = np.arange(1, 13, dtype=np.float64) / 8.0
    # SURVEY 7.0 worked example: y = [35, 30, 7]
    ctl = unit(0, 2, 0, deltas=[2]) + unit(0, 1, 1, nr=True) + unit(1, 3, 0, nr=True)
    cases.append(("survey_worked_example", [export(ctl, [10, 20, 30, 1, 2, 3], [8, 10001], 3, 3)],
                  False, [1.0, 2.0, 3.0], 3, 0.5))
    # delta16 / delta32 units, and a horizontal unit with delta 3
    ctl = (unit(0, 3, 0, deltas=[300, 700], dbytes=2) + unit(1, 2, 5, nr=True, deltas=[70000], dbytes=4)
           + unit(2, 4, 1, nr=True))
    cases.append(("delta16_delta32_horiz3", [export(ctl, [1.5, -2, 3, 4, 5, 6, 7, 8, 9],
                                                    [16, 32, 10003], 3, 70010)],
                  False, {"linspace": [-1.0, 1.0, 70010]}, 3, 1.0))
    # vertical, diagonal, anti-diagonal units scattered below their anchor row
    ctl = (unit(0, 4, 2) + unit(1, 3, 1) + unit(2, 3, 6) + unit(3, 2, 0, nr=True, deltas=[3]))
    cases.append(("vert_diag_rdiag", [export(ctl, [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12],
                                             [20002, 30001, 40002, 8], 8, 12)],
                  False, x12.tolist(), 8, 0.75))
    # block-row 2x3 (column-major values), block-col 3x2 (row-major), br1 1x4, bc1 3x1
    ctl = (unit(0, 6, 1) + unit(1, 6, 3) + unit(2, 4, 2, nr=True, rjmp=3) + unit(3, 3, 0))
    vals = [1, 2, 3, 4, 5, 6, -1, -2, -3, -4, -5, -6, .5, .25, .125, 2.5, 9, 8, 7]
    cases.append(("blocks_and_rowjump", [export(ctl, vals, [60003, 140003, 50004, 130003], 8, 12,
                                                row_jumps=1)],
                  False, x12.tolist(), 8, 2.0))
    # full column indices + a partition that starts at row 4 + leading empty row
    ctl = (unit(0, 2, 3, nr=True, deltas=[4], full=True) + unit(1, 5, 1, nr=True, full=True))
    cases.append(("full_colind_row_start", [export(ctl, [3, 4, 1, 1, 2, 3, 5], [8, 10002], 3, 12,
                                                   row_start=4, full_colind=1)],
                  False, x12.tolist(), 8, -1.0))
    return cases


def from_preprocessor():
    cases = []
    with open(os.path.join(HERE, "reference_matrices.json")) as f:
        mats = json.load(f)
    scen = [
        ("demopatt", {}, False), ("demopatt", {"spx.preproc.xform": "h"}, False),
        ("demopatt", {"spx.preproc.xform": "v"}, False),
        ("demopatt", {"spx.preproc.xform": "all"}, False),
        ("demopatt", {"spx.preproc.xform": "d"}, False),
        ("demopatt", {"spx.preproc.xform": "ad"}, False),
        ("demopatt", {"spx.preproc.xform": "br"}, False),
        ("demopatt", {"spx.preproc.xform": "bc"}, False),
        ("demopatt", {"spx.preproc.xform": "all", "spx.matrix.full_colind": "true"}, False),
        ("demopatt", {"spx.rt.nr_threads": "2", "spx.rt.cpu_affinity": "0,1",
                      "spx.preproc.xform": "all"}, False),
        ("demopatt", {"spx.rt.nr_threads": "2", "spx.preproc.xform": "all",
                      "spx.preproc.sampling.nr_samples": "1",
                      "spx.preproc.sampling.portion": "0.4"}, False),
        ("test", {"spx.preproc.sampling": "none"}, False),
        ("test2", {"spx.preproc.sampling": "none"}, False),
        ("test3", {"spx.preproc.sampling": "none", "spx.rt.nr_threads": "3"}, False),
        ("symmetric", {"spx.preproc.xform": "all"}, True),
        ("symmetric-very-sparse", {"spx.preproc.xform": "all"}, True),
        ("symmetric", {"spx.preproc.xform": "all", "spx.preproc.sampling": "portion",
                       "spx.preproc.sampling.nr_samples": "2",
                       "spx.preproc.sampling.portion": "0.4"}, True),
        ("symmetric", {"spx.rt.nr_threads": "2", "spx.rt.cpu_affinity": "0,1",
                       "spx.preproc.xform": "all"}, True),
        ("symmetric", {"spx.rt.nr_threads": "2", "spx.preproc.xform": "h{1},v{1},d{1}",
                       "spx.matrix.min_unit_size": "2"}, True),
        ("test2", {"spx.preproc.sampling": "none", "spx.rt.nr_threads": "2"}, True),
    ]
    for name, opts, sym in scen:
        m = mats[name]
        sx.options_reset()
        sx.option_set("spx.rt.host_only", "true")
        for k, v in opts.items():
            sx.option_set(k, v)
        if sym:
            sx.option_set("spx.matrix.symmetric", "true")
        inp = sx.input_load_csr(np.array(m["rowptr"], dtype=np.int32),
                                np.array(m["colind"], dtype=np.int32), np.array(m["values"]),
                                m["n"], m["n"])
        A = sx.mat_tune(inp)
        ex = [A.export_csx(p) for p in range(A.info().nr_partitions)]
        x = np.random.RandomState(42).uniform(-0.1, 0.1, m["n"])
        tag = name + "|" + ",".join("%s=%s" % kv for kv in sorted(opts.items())) + ("|sym" if sym else "")
        cases.append((tag, ex, sym, x.tolist(), m["n"], 0.5))
    return cases


def main():
    out = []
    for tag, ex, sym, x, nrows, alpha in hand_built() + from_preprocessor():
        xv = np.linspace(*x["linspace"]) if isinstance(x, dict) else np.array(x)
        y = pyoracle.ref_matvec(ex, sym, xv, nrows, alpha)
        assert y is not None, "reference templates unavailable"
        parts = []
        for e in ex:
            parts.append({"values": e["values"].tolist(), "ctl": e["ctl"].tolist(),
                          "nnz": int(e["nnz"]), "ncols": int(e["ncols"]), "nrows": int(e["nrows"]),
                          "row_start": int(e["row_start"]), "row_jumps": int(e["row_jumps"]),
                          "full_colind": int(e["full_colind"]),
                          "id_map": [int(i) for i in e["id_map"] if i >= 0],
                          "dvalues": None if e["dvalues"] is None else np.asarray(e["dvalues"]).tolist()})
        out.append({"tag": tag, "symmetric": bool(sym), "nrows": nrows, "alpha": alpha, "x": x,
                    "parts": parts, "y_reference_templates": [float(v).hex() for v in y]})
        print("%-70s parts=%d y[:3]=%s" % (tag[:70], len(parts), y[:3]))
    with open(os.path.join(HERE, "csx_vectors.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
