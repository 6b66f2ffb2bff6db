"""The preprocessor's large arrays live in blocks of their own, marked for transparent huge pages
(csrc/big_alloc.hpp): blocks of 32 MB and more, i.e. none of the small matrices of the other CPU
tests.  Here a matrix large enough to take that path is tuned host-only with the marking on and
off (SPX_NO_HUGE_PAGES) in fresh processes: the saved streams must be the same bytes, and the
stream of this process must decode to the matrix."""
import hashlib
import os
import subprocess
import sys

import numpy as np

import sparsex_amd as sx
from sparsex_amd import synth
from helpers import tune
from stream_decode import Stream

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import hashlib, sys, tempfile
sys.path.insert(0, %r)
sys.path.insert(0, %r)
from sparsex_amd import synth
from helpers import tune
csr = synth.syn_nlpkkt_rows(40)
for sym in (False, True):
    A = tune(csr, {"spx.rt.nr_threads": "2"}, sym=sym, host_only=True)
    with tempfile.NamedTemporaryFile(suffix=".spx") as f:
        A.save(f.name)
        print(hashlib.sha256(open(f.name, "rb").read()).hexdigest())
    A.destroy()
""" % (ROOT, os.path.join(ROOT, "tests"))


def hashes(env_extra):
    env = dict(os.environ)
    env.pop("SPX_NO_HUGE_PAGES", None)
    env.update(env_extra)
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return [l for l in out.stdout.split() if len(l) == 64]


def test_large_blocks_give_the_same_stream_with_and_without_huge_pages():
    on, off = hashes({}), hashes({"SPX_NO_HUGE_PAGES": "1"})
    assert len(on) == 2 and on == off


def test_matrix_on_the_large_block_path_decodes_to_itself(tmp_path):
    import scipy.sparse as sp
    csr = synth.syn_nlpkkt_rows(40)
    rp, ci, va, n = csr
    assert 32 * int(rp[-1]) >= (32 << 20)        # 32-byte elements: the element array takes the mapped path
    A = tune(csr, {"spx.rt.nr_threads": "2"}, host_only=True)
    f = str(tmp_path / "m.spx")
    A.save(f)
    s = Stream(f)
    assert s.nnz_stored == rp[-1]
    x = synth.random_x(n)
    a = sp.csr_matrix((va, ci, rp), shape=(n, n))
    assert np.allclose(s.matvec(x), a @ x, rtol=1e-12, atol=1e-14)
    A.destroy()
