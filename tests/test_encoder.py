"""Host preprocessor (mining + encoding + emitters), no GPU needed.

The reference pins none of this with golden outputs, so
the checks are: the two byte streams hand-derived in SURVEY.md section 7.0 from
the reference's code paths, structural invariants, and agreement of the
emitted streams with the CSR product through the oracle.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import sparsex_amd as sx
from sparsex_amd import synth
from helpers import GOLDEN, tune, oracle_y, check_y
from oracle import pyoracle


def _mat(name):
    with open(os.path.join(GOLDEN, "reference_matrices.json")) as f:
        m = json.load(f)[name]
    return (np.array(m["rowptr"], dtype=np.int32), np.array(m["colind"], dtype=np.int32),
            np.array(m["values"]), m["n"])


SURVEY_H = ("00 04 00 01 01 05 81 01 07 81 04 00 01 05 03 81 05 00 01 02 02 04 81 03 00 01 08 "
            "81 04 00 01 04 04 81 02 02 01 80 04 02 01 01 02 80 04 02 80 04 02 01 01 04")
SURVEY_V = ("00 05 00 01 01 01 05 80 01 07 81 04 00 01 04 01 00 01 05 01 04 03 80 02 03 02 "
            "C0 01 02 05 81 04 02 01 04 01 80 03 04 01 02 80 02 04 01 80 03 04 01 04")


@pytest.mark.parametrize("xf,expected,ids", [("h", SURVEY_H, [10001, 8]),
                                             ("v", SURVEY_V, [8, 20001])])
def test_demopatt_ctl_streams_of_survey(xf, expected, ids):
    A = tune(_mat("demopatt"), {"spx.preproc.xform": xf}, host_only=True)
    ex = A.export_csx(0)
    assert " ".join("%02X" % b for b in ex["ctl"]) == expected
    assert [i for i in ex["id_map"] if i >= 0] == ids
    if xf == "h":     # values stay in file order when only in-row units exist
        assert np.array_equal(ex["values"], _mat("demopatt")[2])
    if xf == "v":
        assert ex["values"][:12].tolist() == [1, 2, 3, 4, 5, 6, 7, 11, 16, 19, 8, 12]
        assert ex["row_jumps"] == 1


def test_demopatt_stats_of_survey():
    A = tune(_mat("demopatt"), {"spx.preproc.xform": "h"}, host_only=True)
    log = A.tune_log()
    assert "h:[nz:16, p:4, d:0]: { 1:[nz:16, p:4, d:0] }" in log
    assert "Encode to Horizontal" in log
    A = tune(_mat("demopatt"), {"spx.preproc.xform": "v"}, host_only=True)
    assert "v:[nz:20, p:5, d:0]: { 1:[nz:20, p:5, d:0] }" in A.tune_log()


def _cover(units, n):
    """Expands unit records into the set of (row, col) they cover."""
    seen = {}
    for (t, d, size, r, c) in units:
        for k in range(size):
            if t == 0:
                rc = (r, c)
            elif t == 1:
                rc = (r, c + k * d)
            elif t == 2:
                rc = (r + k * d, c)
            elif t == 3:
                rc = (r + k * d, c + k * d)
            elif t == 4:
                rc = (r + k * d, c - k * d)
            elif 5 <= t <= 12:
                R = t - 4
                rc = (r + k % R, c + k // R)
            else:
                Cc = t - 12
                rc = (r + k // Cc, c + k % Cc)
            assert rc not in seen, "nonzero covered twice: %s" % (rc,)
            seen[rc] = True
    return seen


@pytest.mark.parametrize("gen,opts", [
    (lambda: synth.syn_cant(0.04), {"spx.preproc.sampling": "none"}),
    (lambda: synth.syn_nd24k(0.02), {"spx.preproc.sampling": "none"}),
    (lambda: synth.syn_nlpkkt(7), {"spx.preproc.sampling": "none"}),
    (lambda: synth.syn_nlpkkt(7), {"spx.preproc.sampling": "none", "spx.preproc.heuristic": "cost"}),
    (lambda: synth.syn_nlpkkt(7), {"spx.preproc.sampling": "none", "spx.matrix.split_blocks": "false"}),
    (lambda: synth.syn_webbase(0.01), {"spx.preproc.sampling": "none", "spx.rt.nr_threads": "3"}),
    (lambda: synth.syn_cant(0.04), {"spx.preproc.sampling": "window",
                                    "spx.preproc.sampling.window_size": "400",
                                    "spx.preproc.sampling.nr_samples": "6"}),
    (lambda: synth.syn_cant(0.04), {"spx.preproc.xform": "d{3},bc3{3,13},h{1}"}),
])
def test_every_nonzero_exactly_once_and_product_matches(gen, opts):
    csr = gen()
    rp, ci, va, n = csr
    A = tune(csr, opts, host_only=True)
    inf = A.info()
    total = 0
    rs, re = A.partition()
    for p in range(inf.nr_partitions):
        units = A.export_units(p)
        cov = _cover(units, n)
        total += len(cov)
        # all covered coordinates are nonzeros of the input, in the partition's rows
        for (r, c) in list(cov)[:2000]:
            g = rs[p] + r - 1
            assert rs[p] <= g < max(re[p], rs[p] + 1) or g < n
            cols = ci[rp[g]:rp[g + 1]]
            assert (c - 1) in cols
        min_unit = int(opts.get("spx.matrix.min_unit_size", 4))
        assert all(u[2] >= min_unit for u in units if u[0] != 0)
        assert all(u[2] <= 255 for u in units)
    assert total == rp[-1]
    x = synth.random_x(n)
    yo, ex = oracle_y(A, x, 0.5)
    check_y(csr, x, yo, 0.5)
    assert inf.nnz_stored == rp[-1] and inf.n_unit_elems + inf.n_delta_elems == rp[-1]


def test_symmetric_partitions_and_product():
    csr = synth.syn_cant(0.04)
    rp, ci, va, n = csr
    lower = synth.lower_plus_diag_nnz(rp, ci) - n
    # default: dense 8x8 tiles once, the rest and its mirror image (plus the odd explicit
    # zero on the diagonal that joins two runs)
    B = tune(csr, {"spx.preproc.sampling": "none", "spx.rt.nr_threads": "3"}, sym=True, host_only=True)
    assert lower <= B.info().nnz_stored <= 2 * lower + n
    A = tune(csr, {"spx.preproc.sampling": "none", "spx.rt.nr_threads": "3", "spx.gpu.sym_once": "false"},
             sym=True, host_only=True)
    inf = A.info()
    assert inf.symmetric == 1
    # mirrored stream: HBM holds the strictly lower triangle and its mirror image
    assert inf.nnz_stored == 2 * lower
    x = synth.random_x(n)
    yo, ex = oracle_y(A, x, 0.5)
    check_y(csr, x, yo, 0.5)
    # no unit straddles the partition's first column (CsxManager.hpp:559)
    rs, _ = A.partition()
    for p in range(3):
        for (t, d, size, r, c) in A.export_units(p):
            if t == 1 and size > 1:
                first, last = c, c + (size - 1) * d
                assert not (first <= rs[p] < last)


def test_xform_round_trips():
    """Xform o RevXform = identity for every iteration order
    (cf. the reference's test/src/ElementTest.cpp:61-74)."""
    L = sx.lib()
    L.spx_hip_xform.restype = None
    L.spx_hip_xform.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                C.c_int, C.c_int]
    nr, nc = 13, 17
    for t in range(1, 21):
        for r in range(1, nr + 1):
            for c in range(1, nc + 1):
                rr, cc = C.c_int(r), C.c_int(c)
                L.spx_hip_xform(1, t, C.byref(rr), C.byref(cc), nr, nc)
                assert rr.value >= 1 and cc.value >= 1
                L.spx_hip_xform(t, 1, C.byref(rr), C.byref(cc), nr, nc)
                assert (rr.value, cc.value) == (r, c)
    # spot values: diagonal and block-row maps (Xform.hpp:103-110, 196-203)
    rr, cc = C.c_int(5), C.c_int(7)
    L.spx_hip_xform(1, 3, C.byref(rr), C.byref(cc), nr, nc)
    assert (rr.value, cc.value) == (nr + 7 - 5, 5)
    rr, cc = C.c_int(5), C.c_int(7)
    L.spx_hip_xform(1, 7, C.byref(rr), C.byref(cc), nr, nc)      # br3
    assert (rr.value, cc.value) == ((5 - 1) // 3 + 1, (5 - 1) % 3 + 3 * (7 - 1) + 1)


def test_partition_bounds_balance_by_nnz():
    csr = synth.syn_webbase(0.01)
    rp = csr[0]
    A = tune(csr, {"spx.rt.nr_threads": "4"}, host_only=True)
    rs, re = A.partition()
    assert rs[0] == 0 and all(re[i] == rs[i + 1] for i in range(3)) and re[3] == csr[3]
    # SparseInternal.hpp:131-144: partition i closes at the first row boundary
    # with at least (nnz - taken) / (P - i) elements
    taken = 0
    for i in range(3):
        limit = (rp[-1] - taken) // (4 - i)
        got = rp[re[i]] - rp[rs[i]]
        assert got >= limit
        assert rp[re[i] - 1] - rp[rs[i]] < limit or re[i] - rs[i] == 1
        taken += got
