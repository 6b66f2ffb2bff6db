"""spx_partition_csr (reference src/api/matvec.c:689-737): the split of a CSR matrix over threads.  The library
cuts the row pointer with a search per split; here the rule is restated the slow way -- a running count, row by
row -- and both must name the same splits, on random row pointers with empty rows, heavy rows, more threads than
nonzeros (quota 0) and one- and zero-based pointers."""
import ctypes as C

import numpy as np
import pytest

import sparsex_amd as sx


def splits_the_slow_way(rowptr, nr_rows, nthreads):
    quota = (int(rowptr[nr_rows]) - 1) // nthreads
    rs, re = {0: 0}, {}
    count, k = 0, 0
    for i in range(nr_rows):
        count += int(rowptr[i + 1]) - int(rowptr[i])
        if count >= quota and k < nthreads:
            re[k] = i + 1
            count = 0
            k += 1
            if k < nthreads:
                rs[k] = i + 1
    if count < quota and k < nthreads:
        re[k] = nr_rows + 1                  # (the reference's own: one past the last row)
    return rs, re


def call(rowptr, nr_rows, nthreads):
    L = sx.lib()
    L.spx_partition_csr.restype = C.c_void_p
    L.spx_partition_csr.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    L.spx_partition_get_rs.restype = C.POINTER(C.c_int)
    L.spx_partition_get_rs.argtypes = [C.c_void_p]
    L.spx_partition_get_re.restype = C.POINTER(C.c_int)
    L.spx_partition_get_re.argtypes = [C.c_void_p]
    L.spx_partition_destroy.argtypes = [C.c_void_p]
    rp = np.ascontiguousarray(rowptr, dtype=np.int32)
    p = L.spx_partition_csr(rp.ctypes.data_as(C.c_void_p), nr_rows, nthreads)
    assert p
    rs = [L.spx_partition_get_rs(p)[k] for k in range(nthreads)]
    re = [L.spx_partition_get_re(p)[k] for k in range(nthreads)]
    L.spx_partition_destroy(p)
    return rs, re


@pytest.mark.parametrize("seed", range(40))
def test_same_splits_as_the_running_count(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 400))
    lens = rng.integers(0, 12, size=n)
    if seed % 3 == 0:
        lens[rng.integers(0, n, size=max(1, n // 10))] = 0            # empty rows
    if seed % 4 == 0:
        lens[rng.integers(0, n)] = 5000                               # one heavy row
    if seed % 7 == 0:
        lens[:] = 0
        lens[rng.integers(0, n)] = 3                                  # hardly any nonzeros: quota 0 with many threads
    base = 1 if seed % 2 else 0
    rowptr = np.concatenate([[0], np.cumsum(lens)]) + base
    for t in (1, 2, 3, 7, 16, 64):
        want_rs, want_re = splits_the_slow_way(rowptr, n, t)
        rs, re = call(rowptr, n, t)
        for k, v in want_rs.items():
            assert rs[k] == v, (seed, t, k)
        for k, v in want_re.items():
            assert re[k] == v, (seed, t, k)
