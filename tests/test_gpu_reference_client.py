"""The reference's own test client -- test/src/sparsex_test.c, compiled
unmodified against this repository's headers and libsparsex.so by
oracle/build_ref.py::build_test_client() -- run through the reference's scenario
list (test/scripts/test-sparsex.sh.in:55-244) on the GPU.  The client loads the
matrix with spx_input_load_mmf, tunes, multiplies 128 times with
spx_matvec_mult(0.5, A, x, y) on random host vectors and checks against a serial
product (tests/ref_client/check_result.c, the reference's 1e-6 criterion)."""
import json
import os
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

EXE = os.path.join(ROOT, "oracle", "_ref", "sparsex_test")

SYM = ["-o", "spx.matrix.symmetric=true"]
ALL = ["-o", "spx.preproc.xform=all"]
PORTION = ["-o", "spx.preproc.sampling=portion", "-o", "spx.preproc.sampling.nr_samples=2",
           "-o", "spx.preproc.sampling.portion=0.4"]
TWO = ["-o", "spx.rt.nr_threads=2", "-o", "spx.rt.cpu_affinity=0,1"]
ONE_SAMPLE = ["-o", "spx.preproc.sampling.nr_samples=1", "-o", "spx.preproc.sampling.portion=0.4"]
SCENARIOS = [
    ("demopatt", []),
    ("demopatt", ["-o", "spx.preproc.xform=h"]),
    ("demopatt", ["-o", "spx.preproc.xform=v"]),
    ("demopatt", ALL),
    ("symmetric", ALL + SYM),
    ("symmetric-very-sparse", ALL + SYM),
    ("symmetric", ALL + PORTION + SYM),
    ("symmetric", ALL + PORTION),
    ("demopatt", TWO + ALL),
    ("symmetric", TWO + ALL + SYM),
    ("demopatt", TWO + ALL + ONE_SAMPLE),
    ("symmetric", TWO + ALL + ONE_SAMPLE + SYM),
    ("demopatt", ["-r"] + ALL),                      # reordering (RCM)
    ("symmetric", ["-r"] + ALL + SYM),
    ("demopatt", ["-t"]),                            # timing output
]


def write_sorted_mtx(path, m):
    rp, ci, va, n = m["rowptr"], m["colind"], m["values"], m["n"]
    with open(path, "w") as f:
        f.write("%d %d %d\n" % (n, n, len(va)))
        for r in range(n):
            for k in range(rp[r], rp[r + 1]):
                f.write("%d %d %.17g\n" % (r + 1, ci[k] + 1, va[k]))


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/sparsex_test not built (needs the reference tree at build time)")
@pytest.mark.parametrize("name,args", SCENARIOS, ids=["%s %s" % (n, " ".join(a)) for n, a in SCENARIOS])
def test_reference_client_scenario(tmp_path, name, args):
    with open(os.path.join(GOLDEN, "reference_matrices.json")) as f:
        mats = json.load(f)
    mtx = str(tmp_path / (name + ".mtx.sorted"))
    write_sorted_mtx(mtx, mats[name])
    p = subprocess.run([EXE] + args + [mtx], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0, out[-2000:]
    assert "Check Passed" in out
    if "-t" in args:
        assert "MFLOPS" in out


EXAMPLES = os.path.join(ROOT, "oracle", "_ref", "examples")


@pytest.mark.skipif(not os.path.exists(os.path.join(EXAMPLES, "csr_example")),
                    reason="oracle/_ref/examples not built (needs the reference tree at build time)")
def test_reference_examples_run_unchanged(tmp_path):
    """All six example programs of the reference (src/examples/*.c), compiled unmodified
    against this library by oracle/build_ref.py::build_examples(), run on the GPU: CSR input,
    MMF input, the alpha/beta kernel on tuned buffers, save in one process and restore in
    another (matrix_caching_example_p1/p2), and RCM reordering with vector (inverse)
    reordering.  Each runs its 128 SpMV loops and prints its timing lines."""
    with open(os.path.join(GOLDEN, "reference_matrices.json")) as f:
        mats = json.load(f)
    mtx = str(tmp_path / "demopatt.mtx.sorted")
    write_sorted_mtx(mtx, mats["demopatt"])
    binfile = str(tmp_path / "csx_file.bin")
    runs = [("csr_example", []), ("mmf_example", [mtx]), ("advanced_example", [mtx]),
            ("matrix_caching_example_p1", [mtx, binfile]), ("matrix_caching_example_p2", [binfile]),
            ("reordering_example", [mtx])]
    for name, args in runs:
        p = subprocess.run([os.path.join(EXAMPLES, name)] + args, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, timeout=300, cwd=str(tmp_path))
        out = p.stdout.decode(errors="replace")
        assert p.returncode == 0, name + ": " + out[-2000:]
        assert "SPMV time:" in out and "MFLOPS:" in out, name + ": " + out[-2000:]
        if name == "matrix_caching_example_p1":
            assert os.path.getsize(binfile) > 0
