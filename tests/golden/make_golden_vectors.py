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


class SymStream:
    """Builds the ctl/values of one symmetric partition (lower triangle) unit by unit, tracking
    the column cursor the way the reference's templates move it (delta and horizontal units leave
    it on their last element, everything else where it was: *_sym_tmpl.c), and keeps the
    coordinates for a dense cross-check."""

    def __init__(self, id_map, nrows, ncols, row_start=0, row_jumps=False, dvalues=None):
        self.ids, self.nrows, self.ncols, self.row_start = list(id_map), nrows, ncols, row_start
        self.row_jumps = row_jumps
        self.ctl, self.vals, self.coo = [], [], []
        self.row, self.x = row_start, 0
        self.first = True
        self.dvalues = list(dvalues)

    def _head(self, pid, size, row, col):
        slot = self.ids.index(pid)
        nr, rj = False, 0
        if row != self.row:
            assert row > self.row
            nr = True
            if row - self.row > 1:
                # (a partition with an empty row is emitted with row jumps: the flag only
                # changes the new-row hook, units without the RJMP bit still move one row)
                self.row_jumps = True
                rj = row - self.row
            self.row, self.x = row, 0
        assert col >= self.x
        b = [slot | (0x80 if nr else 0) | (0x40 if rj else 0), size]
        if rj:
            b += varint(rj)
        b += varint(col - self.x)
        self.x = col
        self.ctl += b

    def delta(self, bits, row, cols, vals):
        self._head(bits, len(cols), row, cols[0])
        for a, b in zip(cols, cols[1:]):
            self.ctl += list(int(b - a).to_bytes(bits // 8, "little"))
        self.x = cols[-1]
        self._put([(row, c) for c in cols], vals)

    def horiz(self, d, row, col, vals):
        self._head(10000 + d, len(vals), row, col)
        self.x = col + d * (len(vals) - 1)
        self._put([(row, col + d * i) for i in range(len(vals))], vals)

    def vert(self, d, row, col, vals):
        self._head(20000 + d, len(vals), row, col)
        self._put([(row + d * i, col) for i in range(len(vals))], vals)

    def diag(self, d, row, col, vals):
        self._head(30000 + d, len(vals), row, col)
        self._put([(row + d * i, col + d * i) for i in range(len(vals))], vals)

    def rdiag(self, d, row, col, vals):
        self._head(40000 + d, len(vals), row, col)
        self._put([(row + d * i, col - d * i) for i in range(len(vals))], vals)

    def block_row(self, r, c, row, col, vals):        # values column-major
        self._head((4 + r) * 10000 + c, r * c, row, col)
        self._put([(row + j, col + i) for i in range(c) for j in range(r)], vals)

    def block_col(self, r, c, row, col, vals):        # values row-major
        self._head((12 + c) * 10000 + r, r * c, row, col)
        self._put([(row + i, col + j) for i in range(r) for j in range(c)], vals)

    def _put(self, coords, vals):
        assert len(coords) == len(vals)
        for (r, c), v in zip(coords, vals):
            assert c < r < self.row_start + self.nrows, "unit leaves the strictly lower triangle of its partition"
            self.coo.append((r, c, float(v)))
        self.vals += [float(v) for v in vals]

    def export(self):
        return export(self.ctl, self.vals, self.ids, self.nrows, self.ncols, row_start=self.row_start,
                      row_jumps=1 if self.row_jumps else 0, dvalues=self.dvalues)


def sym_hand_built():
    """Symmetric streams per unit type, built by hand ("the _sym variants with a
    non-NULL local buffer"): vertical, diagonal, anti-diagonal, block-row (r > 2 and r = 1),
    block-col (and c = 1), delta16 / delta32, symmetric row jumps, and partitions with
    row_start > 0 whose units start on either side of it (the reduction target switches between
    the local buffer and y, csx_sym_spmv_tmpl.c:84-95)."""
    cases = []
    n = 12
    x12 = (np.arange(1, 13, dtype=np.float64) / 8.0 - 0.6).tolist()
    dv = [2.0 + 0.25 * i for i in range(n)]

    def one(tag, s, x=x12, alpha=0.5, nrows=n):
        cases.append((tag, [s.export()] if isinstance(s, SymStream) else [t.export() for t in s], True, x,
                      nrows, alpha, s if isinstance(s, list) else [s]))

    s = SymStream([20002, 20001], n, n, dvalues=dv)
    s.vert(2, 1, 0, [1.5, -2.0, 3.0, 0.5])             # rows 1,3,5,7 of column 0
    s.vert(1, 2, 1, [4.0, 5.0])                        # rows 2,3 of column 1
    s.vert(1, 8, 2, [-1.0, 0.25, 7.0, 9.0])            # rows 8..11 of column 2
    one("sym_vert2_vert1", s)

    s = SymStream([30001, 30002], n, n, dvalues=dv)
    s.diag(1, 1, 0, [1.0, 2.0, 3.0, 4.0, 5.0])         # (1,0) ... (5,4)
    s.diag(2, 3, 0, [-1.5, 2.5, 0.125])                # (3,0) (5,2) (7,4)
    s.diag(1, 9, 3, [6.0, -7.0, 8.0])                  # (9,3) (10,4) (11,5)
    one("sym_diag1_diag2", s, alpha=0.75)

    s = SymStream([40001, 40002], n, n, dvalues=dv)
    s.rdiag(1, 4, 3, [1.0, -2.0, 3.0])                 # (4,3) (5,2) (6,1)
    s.rdiag(2, 6, 5, [0.5, 0.25, -0.125])              # (6,5) (8,3) (10,1)
    s.rdiag(1, 9, 8, [9.0, 10.0, 11.0])                # (9,8) (10,7) (11,6)
    one("sym_rdiag1_rdiag2", s, alpha=-1.0)

    s = SymStream([70002, 50004, 80003], n, n, dvalues=dv)
    s.block_row(3, 2, 4, 0, [1, 2, 3, 4, 5, 6])        # rows 4-6 x cols 0-1
    s.block_row(1, 4, 7, 2, [-1, -2, -3, -4])          # row 7 x cols 2-5
    s.block_row(4, 3, 8, 3, [.5, 1.5, 2.5, 3.5, -.5, -1.5, -2.5, -3.5, 8, 7, 6, 5])   # rows 8-11 x cols 3-5
    one("sym_block_row_3x2_1x4_4x3", s, alpha=2.0)

    s = SymStream([150002, 130003, 140004], n, n, dvalues=dv)
    s.block_col(2, 3, 3, 0, [1, 2, 3, 4, 5, 6])        # rows 3-4 x cols 0-2
    s.block_col(3, 1, 5, 1, [-1, -2, -3])              # rows 5-7 x col 1
    s.block_col(4, 2, 8, 4, [.5, 1.5, 2.5, 3.5, -.5, -1.5, -2.5, -3.5])              # rows 8-11 x cols 4-5
    one("sym_block_col_2x3_3x1_4x2", s, alpha=0.5)

    big = 70010
    xb = {"linspace": [-1.0, 1.0, big]}
    s = SymStream([16, 32, 8], big, big, row_jumps=True, dvalues=[1.0 + (i % 7) for i in range(big)])
    s.delta(8, 3, [0, 2], [0.5, 1.5])
    s.delta(16, 400, [10, 310, 399], [1.5, -2.0, 3.0])
    s.delta(32, 70009, [5, 70005], [4.0, 5.0])
    one("sym_delta16_delta32_rowjumps", s, x=xb, alpha=1.0, nrows=big)

    s = SymStream([8, 10002, 20001, 30001, 40001], n, n, row_jumps=True, dvalues=dv)
    s.delta(8, 2, [0, 1], [1.0, 2.0])
    s.horiz(2, 5, 0, [3.0, 4.0, 5.0])                  # (5,0) (5,2) (5,4)
    s.vert(1, 6, 0, [6.0, 7.0])                        # (6,0) (7,0)
    s.diag(1, 6, 2, [8.0, 9.0, 10.0])                  # (6,2) (7,3) (8,4)
    s.rdiag(1, 9, 6, [-1.0, -2.0, -3.0])               # (9,6) (10,5) (11,4)
    s.delta(8, 11, [0, 7, 10], [0.5, 0.25, 0.125])
    one("sym_all_linear_types_rowjumps", s, alpha=0.75)

    # two partitions; the second starts at row 6: its units start on either side of column 6
    p0 = SymStream([8, 10001], 6, n, dvalues=dv[:6])
    p0.delta(8, 1, [0], [1.0])
    p0.horiz(1, 4, 1, [2.0, 3.0, 4.0])
    p1 = SymStream([10001, 8, 20001, 30001], 6, n, row_start=6, dvalues=dv[6:])
    p1.horiz(1, 8, 4, [1.5, 2.5, 3.5])                 # starts in front of row_start: local buffer, also for (8,6)
    p1.delta(8, 8, [7], [4.5])                         # same row, at row_start or beyond: y
    p1.delta(8, 9, [7, 8], [5.5, 6.5])                 # a new row switches back to the buffer, the unit to y
    p1.vert(1, 10, 2, [7.5, 8.5])                      # buffer
    p1.diag(1, 10, 8, [9.5, 10.5])                     # y
    one("sym_two_partitions_buffer_and_y", [p0, p1], alpha=0.5)

    p0 = SymStream([70002], 5, n, row_jumps=True, dvalues=dv[:5])
    p0.block_row(3, 2, 2, 0, [1, 2, 3, 4, 5, 6])
    p1 = SymStream([150002, 70002, 40001], 7, n, row_start=5, row_jumps=True, dvalues=dv[5:])
    p1.block_col(2, 3, 6, 3, [1, -2, 3, -4, 5, -6])     # rows 6-7 x cols 3-5: starts in front of row_start
    p1.block_row(3, 2, 9, 5, [.5, .25, .125, 2, 4, 8])  # rows 9-11 x cols 5-6: at row_start
    p1.rdiag(1, 9, 8, [7.0, 8.0])                       # (9,8) (10,7)
    one("sym_two_partitions_blocks_rowjumps", [p0, p1], alpha=-0.5)
    return cases


def check_dense(streams, x, nrows, alpha, y):
    """The hand-built symmetric streams against a dense product of the matrix they describe."""
    a = np.zeros((nrows, nrows))
    for s in streams:
        for r, c, v in s.coo:
            assert a[r, c] == 0.0, "two units on (%d, %d)" % (r, c)
            a[r, c] = a[c, r] = v
        for i, d in enumerate(s.dvalues):
            a[s.row_start + i, s.row_start + i] = d
    ref = alpha * (a @ x)
    assert np.allclose(y, ref, rtol=1e-12, atol=1e-12), np.abs(y - ref).max()


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


def packed(v):
    import base64
    import zlib
    return {"f64_zlib_b64": base64.b64encode(zlib.compress(np.asarray(v, dtype="<f8").tobytes(), 9)).decode()}


def main():
    out = []
    for case in hand_built() + sym_hand_built() + from_preprocessor():
        tag, ex, sym, x, nrows, alpha = case[:6]
        xv = np.linspace(*x["linspace"]) if isinstance(x, dict) else np.array(x)
        y = pyoracle.ref_matvec(ex, sym, xv, nrows, alpha)
        assert y is not None, "reference templates unavailable"
        if len(case) > 6 and nrows <= 4096:
            check_dense(case[6], xv, nrows, alpha, y)
        parts = []
        big = nrows > 4096              # long vectors travel as zlib + base64 of their float64 bytes
        for e in ex:
            parts.append({"values": e["values"].tolist(), "ctl": e["ctl"].tolist(),
                          "nnz": int(e["nnz"]), "ncols": int(e["ncols"]), "nrows": int(e["nrows"]),
                          "row_start": int(e["row_start"]), "row_jumps": int(e["row_jumps"]),
                          "full_colind": int(e["full_colind"]),
                          "id_map": [int(i) for i in e["id_map"] if i >= 0],
                          "dvalues": None if e["dvalues"] is None else
                          (packed(e["dvalues"]) if big else np.asarray(e["dvalues"]).tolist())})
        out.append({"tag": tag, "symmetric": bool(sym), "nrows": nrows, "alpha": alpha, "x": x,
                    "parts": parts,
                    "y_reference_templates": packed(y) if big else [float(v).hex() for v in y]})
        print("%-70s parts=%d y[:3]=%s" % (tag[:70], len(parts), y[:3]))
    with open(os.path.join(HERE, "csx_vectors.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
