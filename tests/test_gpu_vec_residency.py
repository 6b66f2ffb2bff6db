"""spx.vec.device (opt-in; default off since round 6): vectors created by the library carry a version that every
spx_vec_* mutator advances; spx_matvec_* reuse x's copy in HBM while the version stands, so
the 128-loop of a relinked reference client (test/src/sparsex_test.c:161-163) uploads x once.
Results must follow every change made through the API."""
import ctypes as C
import time

import numpy as np
import pytest

import sparsex_amd as sx
from sparsex_amd import synth
from sparsex_amd.api import VectorStruct
from helpers import tune, check_y

pytestmark = pytest.mark.gpu


def _vecs(n, part):
    L = sx.lib()
    L.spx_vec_create_random.restype = C.POINTER(VectorStruct)
    L.spx_vec_create_random.argtypes = [C.c_size_t, C.c_void_p]
    L.spx_vec_create.restype = C.POINTER(VectorStruct)
    L.spx_vec_create.argtypes = [C.c_size_t, C.c_void_p]
    x, y = L.spx_vec_create_random(n, part), L.spx_vec_create(n, part)
    xa = np.ctypeslib.as_array(x.contents.elements, shape=(n,))
    ya = np.ctypeslib.as_array(y.contents.elements, shape=(n,))
    return x, y, xa, ya


@pytest.mark.parametrize("resident", ["true", "false"])
def test_x_stays_in_hbm_between_calls(resident):
    csr = synth.syn_cant(0.2)
    n = csr[3]
    A = tune(csr, {"spx.vec.device": resident})
    L = sx.lib()
    L.spx_mat_get_partition.restype = C.c_void_p
    part = C.c_void_p(L.spx_mat_get_partition(C.c_void_p(A.handle)))
    x, y, xa, ya = _vecs(n, part)
    L.spx_vec_scale.argtypes = [C.POINTER(VectorStruct), C.POINTER(VectorStruct), C.c_double]
    L.spx_vec_scale.restype = None
    L.spx_vec_set_entry.argtypes = [C.POINTER(VectorStruct), C.c_int, C.c_double, C.c_int]
    for step in range(3):
        assert L.spx_matvec_mult(0.5, C.c_void_p(A.handle), x, y) == 0
        check_y(csr, xa.copy(), ya.copy(), 0.5)
    L.spx_vec_scale(x, x, 3.0)                                   # a mutator: the next call uploads again
    assert L.spx_matvec_mult(0.5, C.c_void_p(A.handle), x, y) == 0
    check_y(csr, xa.copy(), ya.copy(), 0.5)
    assert L.spx_vec_set_entry(x, 5, 7.25, sx.SPX_INDEX_ZERO_BASED) == 0
    assert xa[5] == 7.25
    assert L.spx_matvec_mult(0.5, C.c_void_p(A.handle), x, y) == 0
    check_y(csr, xa.copy(), ya.copy(), 0.5)
    # a client that rewrites the vector through the public struct, behind the library's back: the fingerprint
    # of the contents sees it and x travels again
    xa[:] = np.cos(np.arange(n))
    assert L.spx_matvec_mult(0.5, C.c_void_p(A.handle), x, y) == 0
    check_y(csr, xa.copy(), ya.copy(), 0.5)
    t0 = time.perf_counter()
    for _ in range(128):
        L.spx_matvec_mult(0.5, C.c_void_p(A.handle), x, y)
    us = (time.perf_counter() - t0) / 128 * 1e6
    print("spx.vec.device=%s: %.1f us per spx_matvec_mult call (n = %d)" % (resident, us, n))
    check_y(csr, xa.copy(), ya.copy(), 0.5)
    L.spx_vec_destroy(x)
    L.spx_vec_destroy(y)
    L.spx_partition_destroy.argtypes = [C.c_void_p]
    L.spx_partition_destroy(part)


@pytest.mark.parametrize("resident", ["default", "true"])
def test_one_element_written_through_the_public_struct(resident):
    """`struct vector_struct` is public (include/sparsex/common.h; reference Vector.hpp:30-35): a client may poke one
    element of a spx_vec_create'd vector -- the unit-vector sweep x[i-1] = 0, x[i] = 1.  With the DEFAULT options
    every call must see it (x travels every time); a client that opted in to resident vectors says so with
    spx_hip_vec_touch."""
    csr = synth.syn_cant(0.1)
    n = csr[3]
    A = tune(csr, {} if resident == "default" else {"spx.vec.device": "true"})
    L = sx.lib()
    L.spx_mat_get_partition.restype = C.c_void_p
    part = C.c_void_p(L.spx_mat_get_partition(C.c_void_p(A.handle)))
    x, y, xa, ya = _vecs(n, part)
    L.spx_hip_vec_touch.argtypes = [C.POINTER(VectorStruct)]
    L.spx_hip_vec_touch.restype = None
    xa[:] = 0.0
    L.spx_hip_vec_touch(x)
    for i in (n // 2, n // 2 + 1, n // 2 + 2, 3):          # (not among the few hundred sampled positions by design)
        if i > 0:
            xa[i - 1] = 0.0
        xa[i] = 1.0
        if resident == "true":
            L.spx_hip_vec_touch(x)
        assert L.spx_matvec_mult(1.0, C.c_void_p(A.handle), x, y) == 0
        check_y(csr, xa.copy(), ya.copy(), 1.0)
        xa[i] = 0.0
    L.spx_vec_destroy(x)
    L.spx_vec_destroy(y)
    L.spx_partition_destroy.argtypes = [C.c_void_p]
    L.spx_partition_destroy(part)
    sx.options_reset()
