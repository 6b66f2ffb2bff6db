"""bench.py's CPU-baseline leg (oracle/cpu_baseline.c): the threaded harness that
restates the reference's execution model -- one partition per pinned thread,
spin barriers; symmetric: local buffers and the conflict-map reduction of
MatVecMult_sym (src/internals/CsxKernels.cpp:105-129, CsxSpmv.cpp:37-50,
CsxBuild.hpp:400-581) -- must compute the product it times."""
import numpy as np
import pytest

import bench
from sparsex_amd import synth
from oracle import pyoracle


@pytest.mark.parametrize("symmetric", [False, True])
@pytest.mark.parametrize("threads", [1, 3, 4])
@pytest.mark.parametrize("gen", ["cant", "nlpkkt"])
def test_threaded_baseline_computes_the_product(gen, threads, symmetric):
    csr = synth.syn_cant(0.04) if gen == "cant" else synth.syn_nlpkkt_rows(8)
    rp, ci, va, n = csr
    x = synth.random_x(n)
    ex = bench.baseline_partitions(csr, threads, symmetric)
    from oracle import build_ref
    if build_ref.reference_available():
        # where the reference tree is mounted the per-partition routine is the reference's
        # own template code (src/templates/csx[_sym]_spmv_tmpl.c + unit bodies), built in place
        for e in ex:
            key = bench._ref_key(e)
            if key[0]:
                build_ref.build(key[0], symmetric, key[1], key[2], opt="-O3")
    sec, kind, y = bench._time_baseline(ex, csr, x, n, threads, symmetric, loops=3, batches=2)
    assert sec > 0 and kind == ("reference" if build_ref.reference_available() else "port")
    yc = bench.ALPHA * pyoracle.csr_matvec(rp, ci, va, x)
    assert pyoracle.vec_compare(yc, y) == 0
    assert np.allclose(y, yc, rtol=1e-11, atol=1e-13)


def test_cuts_follow_the_reference_rule():
    counts = np.array([5, 1, 1, 1, 8, 2, 2, 4, 4, 4])
    cuts = bench.nnz_balanced_cuts(counts, 3)
    assert cuts[0] == 0 and cuts[-1] == counts.size and cuts == sorted(cuts)
    # rank i closes once it holds at least (remaining nonzeros) / (ranks left)
    cum = np.concatenate([[0], np.cumsum(counts)])
    taken = 0
    for i in range(2):
        limit = (cum[-1] - taken) // (3 - i)
        got = cum[cuts[i + 1]] - taken
        assert got >= limit and got - counts[cuts[i + 1] - 1] < limit
        taken = cum[cuts[i + 1]]
