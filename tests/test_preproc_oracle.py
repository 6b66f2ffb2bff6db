"""The product's host preprocessor against the independent Python restatement
(oracle/csx_preproc.py): same partitions, same units, same ctl bytes, on small
inputs under the option combinations the reference's own tests use
(test/scripts/test-sparsex.sh.in:55-244) and more."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN, tune
from oracle import csx_preproc as ora
from sparsex_amd import synth


def _mat(name):
    with open(os.path.join(GOLDEN, "reference_matrices.json")) as f:
        m = json.load(f)[name]
    return (np.array(m["rowptr"], dtype=np.int32), np.array(m["colind"], dtype=np.int32),
            np.array(m["values"]), m["n"])


def _triplets(csr):
    rp, ci, va, n = csr
    return [(r + 1, int(ci[j]) + 1, float(va[j])) for r in range(n) for j in range(rp[r], rp[r + 1])]


CASES = []
for name in ["demopatt", "test", "test2", "test3"]:
    for opts in [{}, {"spx.preproc.xform": "h"}, {"spx.preproc.xform": "v"},
                 {"spx.preproc.xform": "d,ad"}, {"spx.preproc.xform": "br,bc"},
                 {"spx.preproc.sampling": "none"},
                 {"spx.preproc.sampling": "none", "spx.matrix.split_blocks": "false"},
                 {"spx.preproc.sampling": "none", "spx.preproc.heuristic": "cost"},
                 {"spx.preproc.sampling": "none", "spx.matrix.min_unit_size": "2",
                  "spx.matrix.min_coverage": "0.05"},
                 {"spx.rt.nr_threads": "2", "spx.preproc.sampling": "none"},
                 {"spx.rt.nr_threads": "2", "spx.preproc.sampling.nr_samples": "1",
                  "spx.preproc.sampling.portion": "0.4"},
                 {"spx.preproc.xform": "h{1},v{1},d{1,2}", "spx.matrix.min_unit_size": "3"}]:
        CASES.append((name, opts, False))
for name in ["symmetric", "symmetric-very-sparse", "test2"]:
    for opts in [{}, {"spx.preproc.sampling": "none"},
                 {"spx.preproc.sampling": "none", "spx.rt.nr_threads": "2"},
                 {"spx.preproc.sampling": "portion", "spx.preproc.sampling.nr_samples": "2",
                  "spx.preproc.sampling.portion": "0.4"}]:
        CASES.append((name, opts, True))


def _id(c):
    return "%s|%s|%s" % (c[0], ",".join("%s=%s" % (k.split(".")[-1], v) for k, v in c[1].items()),
                         "sym" if c[2] else "gen")


def _compare(csr, opts, sym):
    o = dict(opts)
    if sym:
        o["spx.matrix.symmetric"] = "true"
    A = tune(csr, opts, sym=sym, host_only=True)
    parts = ora.preprocess(_triplets(csr), csr[3], csr[3], o)
    inf = A.info()
    assert inf.nr_partitions == len(parts)
    rs, re = A.partition()
    for p, part in enumerate(parts):
        assert rs[p] == part.row_start
        assert A.export_units(p) == ora.units_of(part)
        ex = A.export_csx(p)
        ctl, values, id_map, row_jumps = ora.emit_ctl(part, symmetric=sym)
        assert bytes(ex["ctl"]) == ctl
        assert np.array_equal(ex["values"], np.array(values))
        assert [i for i in ex["id_map"] if i >= 0] == id_map
        assert bool(ex["row_jumps"]) == row_jumps


@pytest.mark.parametrize("case", CASES, ids=[_id(c) for c in CASES])
def test_reference_fixtures(case):
    name, opts, sym = case
    _compare(_mat(name), opts, sym)


@pytest.mark.parametrize("gen,opts,sym", [
    (lambda: synth.syn_cant(0.012), {"spx.preproc.sampling": "none"}, False),
    (lambda: synth.syn_cant(0.012), {"spx.preproc.sampling": "none", "spx.rt.nr_threads": "3"}, True),
    (lambda: synth.syn_cant(0.02), {"spx.preproc.sampling": "window",
                                    "spx.preproc.sampling.window_size": "300",
                                    "spx.preproc.sampling.nr_samples": "5"}, False),
    (lambda: synth.syn_nd24k(0.012), {"spx.preproc.sampling": "none"}, False),
    (lambda: synth.syn_nd24k(0.012), {"spx.preproc.sampling": "portion",
                                      "spx.preproc.sampling.portion": "0.3",
                                      "spx.preproc.sampling.nr_samples": "6"}, False),
    (lambda: synth.syn_nlpkkt(5), {"spx.preproc.sampling": "none"}, False),
    (lambda: synth.syn_nlpkkt(5), {"spx.preproc.sampling": "none", "spx.rt.nr_threads": "2"}, True),
    (lambda: synth.syn_webbase(0.003), {"spx.preproc.sampling": "none", "spx.rt.nr_threads": "2"}, False),
    (lambda: synth.syn_cant(0.012), {"spx.preproc.sampling": "none",
                                     "spx.matrix.full_colind": "true"}, False),
])
def test_synthetic_small(gen, opts, sym):
    csr = gen()
    if opts.get("spx.matrix.full_colind") == "true":
        # the Python emitter is exercised with full column indices too
        A = tune(csr, opts, sym=sym, host_only=True)
        parts = ora.preprocess(_triplets(csr), csr[3], csr[3], dict(opts))
        ex = A.export_csx(0)
        ctl, values, id_map, _ = ora.emit_ctl(parts[0], full_colind=True)
        assert bytes(ex["ctl"]) == ctl and A.export_units(0) == ora.units_of(parts[0])
        return
    _compare(csr, opts, sym)
