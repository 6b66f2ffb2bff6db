"""Python face of the parity oracle (TEST INFRASTRUCTURE ONLY).

Wraps oracle/libcsx_oracle.so (the C restatement, csx_oracle.c) and, when
present, the reference-template builds under oracle/_ref/ (build_ref.py).
Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never by the sparsex_amd package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import build_ref

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OracleCsx(C.Structure):
    _fields_ = [("values", C.POINTER(C.c_double)), ("ctl", C.POINTER(C.c_uint8)),
                ("ctl_size", C.c_int64), ("nnz", C.c_int32), ("ncols", C.c_int32),
                ("nrows", C.c_int32), ("row_start", C.c_int32), ("row_jumps", C.c_int32),
                ("full_colind", C.c_int32), ("id_map", C.c_long * 64),
                ("dvalues", C.POINTER(C.c_double))]


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(HERE, "libcsx_oracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", HERE, "-s"])
        L = C.CDLL(so)
        dp = C.POINTER(C.c_double)
        L.oracle_csx_multiply.argtypes = [C.POINTER(OracleCsx), dp, dp, C.c_double]
        L.oracle_csx_multiply.restype = None
        L.oracle_csx_sym_multiply.argtypes = [C.POINTER(OracleCsx), dp, dp, dp, C.c_double]
        L.oracle_csx_sym_multiply.restype = None
        L.oracle_csr_spmv.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), dp, dp, dp]
        L.oracle_csr_spmv.restype = None
        L.oracle_vec_compare.argtypes = [dp, dp, C.c_long]
        L.oracle_matvec_mult.argtypes = [C.POINTER(OracleCsx), C.c_int, C.c_int, C.c_long, dp,
                                         dp, C.c_double, C.c_int, dp]
        L.oracle_matvec_mult.restype = None
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Partitions:
    """Exported CSX partitions (dicts from Matrix.export_csx) in oracle form."""

    def __init__(self, exports, symmetric):
        self.exports = exports
        self.symmetric = bool(symmetric)
        self.n = len(exports)
        self.arr = (OracleCsx * self.n)()
        self._keep = []
        for i, e in enumerate(exports):
            vals = np.ascontiguousarray(e["values"], dtype=np.float64)
            ctl = np.ascontiguousarray(e["ctl"], dtype=np.uint8)
            if vals.size == 0:
                vals = np.zeros(1)
            if ctl.size == 0:
                ctl = np.zeros(1, dtype=np.uint8)
            self._keep += [vals, ctl]
            m = self.arr[i]
            m.values = _dp(vals)
            m.ctl = ctl.ctypes.data_as(C.POINTER(C.c_uint8))
            m.ctl_size = int(e["ctl"].size)
            m.nnz, m.ncols, m.nrows = e["nnz"], e["ncols"], e["nrows"]
            m.row_start = e["row_start"]
            m.row_jumps = e["row_jumps"]
            m.full_colind = e["full_colind"]
            for k in range(64):
                m.id_map[k] = e["id_map"][k]
            if self.symmetric:
                dv = np.ascontiguousarray(e["dvalues"], dtype=np.float64)
                if dv.size == 0:
                    dv = np.zeros(1)
                self._keep.append(dv)
                m.dvalues = _dp(dv)


def csx_matvec(parts, x, nrows, alpha=1.0, nthreads=1):
    """y = alpha*A*x through the C restatement of the reference's CSX kernels."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros(nrows)
    scratch = np.zeros(max(1, (parts.n - 1) * nrows if parts.symmetric else 1))
    lib().oracle_matvec_mult(parts.arr, parts.n, int(parts.symmetric), nrows, _dp(x), _dp(y),
                             float(alpha), int(nthreads), _dp(scratch))
    return y


def csr_matvec(rowptr, colind, values, x):
    """The reference tests' ground truth: serial CSR loop (0-based arrays)."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    colind = np.ascontiguousarray(colind, dtype=np.int32)
    values = np.ascontiguousarray(values, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    n = rowptr.size - 1
    y = np.zeros(n)
    lib().oracle_csr_spmv(n, rowptr.ctypes.data_as(C.POINTER(C.c_int)),
                          colind.ctypes.data_as(C.POINTER(C.c_int)), _dp(values), _dp(x), _dp(y))
    return y


def vec_compare(a, b):
    """0 if equal within the reference's relative 1e-6, else 1-based index."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return lib().oracle_vec_compare(_dp(a), _dp(b), a.size)


def ref_matvec(exports, symmetric, x, nrows, alpha=1.0, build=True):
    """y = alpha*A*x through the reference's own templates (oracle/_ref).

    Returns None when no build for the partitions' pattern sets is available.
    """
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros(nrows)
    xin = build_ref.RefVector(_dp(x), x.size, 1, 45)
    yout = build_ref.RefVector(_dp(y), y.size, 1, 45)
    tmps = []
    for p, e in enumerate(exports):
        if e["ctl"].size == 0 and not symmetric:
            continue
        ids = [i for i in e["id_map"] if i >= 0]
        if not ids:
            # a symmetric partition without lower elements: diagonal only
            if symmetric and e["nrows"] > 0:
                rs, nr = e["row_start"], e["nrows"]
                y[rs:rs + nr] += x[rs:rs + nr] * e["dvalues"][:nr] * alpha
            continue
        fn = (build_ref.build if build else build_ref.lookup)(
            ids, symmetric, bool(e["row_jumps"]), bool(e["full_colind"]))
        if fn is None:
            return None
        L = C.CDLL(fn)
        m = build_ref.RefCsxMatrix()
        # (the templates load the value FOLLOWING a unit's last one before they notice the
        # unit has ended, e.g. horiz_sym_tmpl.c:38-39: one element of slack, never used)
        vals = np.concatenate([np.asarray(e["values"], dtype=np.float64), [0.0]])
        ctl = np.ascontiguousarray(e["ctl"], dtype=np.uint8)
        m.values = _dp(vals)
        m.ctl = ctl.ctypes.data_as(C.POINTER(C.c_uint8))
        m.nnz, m.ncols, m.nrows = e["nnz"], e["ncols"], e["nrows"]
        m.ctl_size = int(ctl.size)
        m.row_start = e["row_start"]
        m.row_jumps = e["row_jumps"]
        for k in range(63):
            m.id_map[k] = e["id_map"][k]
        if not symmetric:
            f = L.spm_csx_multiply
            f.restype = None
            f.argtypes = [C.c_void_p, C.POINTER(build_ref.RefVector),
                          C.POINTER(build_ref.RefVector), C.c_double, C.c_void_p]
            f(C.byref(m), C.byref(xin), C.byref(yout), float(alpha), None)
        else:
            sm = build_ref.RefCsxSymMatrix()
            sm.lower_matrix = C.pointer(m)
            dv = np.ascontiguousarray(e["dvalues"], dtype=np.float64)
            sm.dvalues = _dp(dv)
            if p == 0:
                tmpv = yout
            else:
                t = np.zeros(nrows)
                tmps.append(t)
                tmpv = build_ref.RefVector(_dp(t), t.size, 1, 45)
            f = L.spm_csx_sym_multiply
            f.restype = None
            f.argtypes = [C.c_void_p, C.POINTER(build_ref.RefVector),
                          C.POINTER(build_ref.RefVector), C.c_double,
                          C.POINTER(build_ref.RefVector)]
            f(C.byref(sm), C.byref(xin), C.byref(yout), float(alpha), C.byref(tmpv))
    for t in tmps:
        y += t
    return y
