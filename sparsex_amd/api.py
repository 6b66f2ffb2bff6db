"""ctypes binding of libsparsex.so.

Names follow the C API one to one (``spx_input_load_csr`` ->
:func:`input_load_csr` etc.; reference ``include/sparsex/matvec.h:39-535``).
There is no Python or CPU fallback for the multiplication: if the shared
library (or, at call time, a HIP device) is missing, calls raise.
"""
import ctypes as C
import os

import numpy as np

SPX_SUCCESS = 0
SPX_FAILURE = -1
SPX_MAT_REORDER = 42
SPX_VEC_AS_IS = 43
SPX_VEC_TUNE = 44
SPX_INDEX_ZERO_BASED = 45
SPX_INDEX_ONE_BASED = 46

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    # SPX_LIB_PATH selects an alternative build of the same library (kernel
    # experiments); the default is the in-tree product build
    return os.environ.get("SPX_LIB_PATH") or os.path.join(_HERE, "lib", "libsparsex.so")


class SpxError(RuntimeError):
    pass


class VectorStruct(C.Structure):
    """``struct vector_struct`` -- public ABI (include/sparsex/common.h)."""
    _fields_ = [("elements", C.POINTER(C.c_double)), ("size", C.c_size_t),
                ("alloc_type", C.c_int), ("vec_mode", C.c_int)]


class HipInfo(C.Structure):
    _fields_ = [("nnz", C.c_int64), ("nnz_stored", C.c_int64),
                ("n_unit_elems", C.c_int64), ("n_delta_elems", C.c_int64),
                ("n_units", C.c_int64), ("n_rowblocks", C.c_int64),
                ("n_shared_rows", C.c_int64), ("value_bytes", C.c_int64),
                ("index_bytes", C.c_int64), ("nr_partitions", C.c_int32),
                ("first_partition", C.c_int32), ("last_partition", C.c_int32),
                ("row_lo", C.c_int32), ("row_hi", C.c_int32),
                ("symmetric", C.c_int32), ("on_device", C.c_int32),
                ("device", C.c_int32), ("waves", C.c_int32), ("sym_tiles", C.c_int32),
                ("tune_seconds", C.c_double),
                ("emit_seconds", C.c_double), ("wave_tiles", C.c_int32), ("sym_segments", C.c_int32),
                ("quad", C.c_int32), ("col_slices", C.c_int32),
                ("unit_windows", C.c_int32), ("unit_window_lds", C.c_int32),
                ("unit_window_elems", C.c_int64), ("unit_window_staged", C.c_int64),
                ("sym_pipeline", C.c_int32), ("reserved0", C.c_int32), ("sym_pipeline_elems", C.c_int64)]


class SxPlan(C.Structure):
    _fields_ = [("passes", C.c_void_p), ("n_sx", C.POINTER(C.c_uint32)), ("n_rowblocks", C.c_size_t),
                ("n_passes", C.c_size_t), ("rowblocks_with_sx", C.c_size_t), ("sym_elems", C.c_uint64),
                ("sx_elems", C.c_uint64), ("sym_passes", C.c_uint64), ("sx_passes", C.c_uint64)]


class XwPlan(C.Structure):
    _fields_ = [("tab", C.POINTER(C.c_uint32)), ("xdescs", C.POINTER(C.c_uint32)), ("passes", C.c_void_p),
                ("n_rowblocks", C.c_size_t), ("n_descs", C.c_size_t), ("n_passes", C.c_size_t),
                ("rowblocks_with_windows", C.c_size_t), ("rowblocks_with_units", C.c_size_t),
                ("staged_doubles", C.c_uint64), ("unit_elems", C.c_uint64), ("unit_elems_lds", C.c_uint64),
                ("lds_doubles", C.c_uint32)]


class CsxExport(C.Structure):
    _fields_ = [("values", C.POINTER(C.c_double)), ("ctl", C.POINTER(C.c_uint8)),
                ("ctl_size", C.c_int64), ("nnz", C.c_int), ("ncols", C.c_int),
                ("nrows", C.c_int), ("row_start", C.c_int),
                ("row_jumps", C.c_int32), ("full_colind", C.c_int32),
                ("id_map", C.c_long * 64), ("rows_info", C.POINTER(C.c_int)),
                ("dvalues", C.POINTER(C.c_double))]


class UnitRecord(C.Structure):
    _fields_ = [("type", C.c_int32), ("delta", C.c_int32), ("size", C.c_int32),
                ("row", C.c_int32), ("col", C.c_int32)]


_lib = None


def lib():
    """Loads libsparsex.so (once).  Raises SpxError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise SpxError(
            "libsparsex.so is not built (%s); run `make lib` or "
            "__graft_entry__.build() -- there is no fallback path" % path)
    L = C.CDLL(path, mode=C.RTLD_GLOBAL)
    vp, i, d = C.c_void_p, C.c_int, C.c_double
    L.spx_input_load_csr.restype = vp
    L.spx_input_load_mmf.restype = vp
    L.spx_input_load_mmf.argtypes = [C.c_char_p]
    L.spx_input_destroy.argtypes = [vp]
    L.spx_mat_tune.restype = vp
    L.spx_mat_destroy.argtypes = [vp]
    L.spx_mat_get_nrows.argtypes = [vp]
    L.spx_mat_get_ncols.argtypes = [vp]
    L.spx_mat_get_nnz.argtypes = [vp]
    L.spx_mat_get_partition.restype = vp
    L.spx_mat_get_partition.argtypes = [vp]
    L.spx_partition_get_rs.restype = C.POINTER(C.c_int)
    L.spx_partition_get_rs.argtypes = [vp]
    L.spx_partition_get_re.restype = C.POINTER(C.c_int)
    L.spx_partition_get_re.argtypes = [vp]
    L.spx_partition_destroy.argtypes = [vp]
    L.spx_option_set.argtypes = [C.c_char_p, C.c_char_p]
    L.spx_option_set.restype = None
    L.spx_hip_options_reset.restype = None
    L.spx_vec_create_from_buff.restype = C.POINTER(VectorStruct)
    L.spx_vec_create_from_buff.argtypes = [C.POINTER(C.c_double), vp, C.c_size_t, vp,
                                           C.c_uint]
    L.spx_vec_destroy.argtypes = [C.POINTER(VectorStruct)]
    L.spx_vec_destroy.restype = None
    L.spx_matvec_mult.argtypes = [d, vp, C.POINTER(VectorStruct), C.POINTER(VectorStruct)]
    L.spx_matvec_kernel.argtypes = [d, vp, C.POINTER(VectorStruct), d,
                                    C.POINTER(VectorStruct)]
    L.spx_hip_matvec_mult.argtypes = [d, vp, vp, vp, vp]
    L.spx_hip_matvec_kernel.argtypes = [d, vp, vp, d, vp, vp]
    L.spx_hip_mat_info.argtypes = [vp, C.POINTER(HipInfo)]
    L.spx_hip_mat_export_csx.argtypes = [vp, i, C.POINTER(CsxExport)]
    L.spx_hip_mat_export_units.restype = C.c_int64
    L.spx_hip_mat_export_units.argtypes = [vp, i, C.POINTER(UnitRecord), C.c_int64]
    L.spx_hip_mat_tune_log.restype = C.c_char_p
    L.spx_hip_mat_tune_log.argtypes = [vp]
    L.spx_hip_vec_create.restype = vp
    L.spx_hip_vec_create.argtypes = [C.c_size_t]
    L.spx_hip_vec_destroy.argtypes = [vp]
    L.spx_hip_vec_data.restype = vp
    L.spx_hip_vec_data.argtypes = [vp]
    L.spx_hip_vec_size.restype = C.c_size_t
    L.spx_hip_vec_size.argtypes = [vp]
    L.spx_hip_vec_upload.argtypes = [vp, C.POINTER(VectorStruct), vp]
    L.spx_hip_vec_download.argtypes = [vp, C.POINTER(VectorStruct), vp]
    L.spx_hip_vec_init.argtypes = [vp, d, vp]
    L.spx_hip_vec_scale.argtypes = [vp, vp, d, vp]
    L.spx_hip_vec_scale_add.argtypes = [vp, vp, vp, d, vp]
    L.spx_hip_vec_add.argtypes = [vp, vp, vp, vp]
    L.spx_hip_vec_sub.argtypes = [vp, vp, vp, vp]
    L.spx_hip_vec_mul.argtypes = [vp, vp, C.POINTER(C.c_double), vp]
    L.spx_hip_vec_copy.argtypes = [vp, vp, vp]
    L.spx_hip_matvec_kernel_vec.argtypes = [d, vp, vp, d, vp, vp]
    L.spx_log_disable_all.restype = None
    L.spx_log_error_console.restype = None
    _lib = L
    return L


def option_set(option, value):
    """``spx_option_set(option, value)`` (reference src/api/matvec.c:753-756)."""
    lib().spx_option_set(str(option).encode(), str(value).encode())


def options_reset():
    lib().spx_hip_options_reset()


class Input:
    """``spx_input_t`` handle; keeps the borrowed CSR arrays alive."""

    def __init__(self, handle, keep=()):
        self.handle = handle
        self._keep = keep

    def destroy(self):
        if self.handle:
            lib().spx_input_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def input_load_csr(rowptr, colind, values, nrows, ncols, indexing=SPX_INDEX_ZERO_BASED):
    """``spx_input_load_csr`` -- wraps (does not copy) the CSR arrays."""
    rp = np.ascontiguousarray(rowptr, dtype=np.int32)
    ci = np.ascontiguousarray(colind, dtype=np.int32)
    va = np.ascontiguousarray(values, dtype=np.float64)
    h = lib().spx_input_load_csr(
        rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p),
        va.ctypes.data_as(C.c_void_p), C.c_int(nrows), C.c_int(ncols), C.c_int(indexing))
    if not h:
        raise SpxError("spx_input_load_csr failed")
    return Input(h, keep=(rp, ci, va))


def input_load_mmf(filename):
    """``spx_input_load_mmf``."""
    h = lib().spx_input_load_mmf(str(filename).encode())
    if not h:
        raise SpxError("spx_input_load_mmf failed for %s" % filename)
    return Input(h)


class Matrix:
    """``spx_matrix_t`` handle with the multiplication entry points."""

    def __init__(self, handle):
        self.handle = handle
        L = lib()
        self.nrows = L.spx_mat_get_nrows(handle)
        self.ncols = L.spx_mat_get_ncols(handle)
        self.nnz = L.spx_mat_get_nnz(handle)

    # -- reference API ---------------------------------------------------
    def matvec_mult(self, alpha, x, y):
        """``spx_matvec_mult(alpha, A, x, y)`` on host numpy vectors."""
        return self._host(alpha, x, None, y)

    def matvec_kernel(self, alpha, x, beta, y):
        """``spx_matvec_kernel(alpha, A, x, beta, y)`` on host numpy vectors."""
        return self._host(alpha, x, beta, y)

    def _host(self, alpha, x, beta, y):
        L = lib()
        assert x.dtype == np.float64 and y.dtype == np.float64
        assert x.flags.c_contiguous and y.flags.c_contiguous
        xv = L.spx_vec_create_from_buff(x.ctypes.data_as(C.POINTER(C.c_double)), None,
                                        x.size, None, SPX_VEC_AS_IS)
        yv = L.spx_vec_create_from_buff(y.ctypes.data_as(C.POINTER(C.c_double)), None,
                                        y.size, None, SPX_VEC_AS_IS)
        try:
            if beta is None:
                rc = L.spx_matvec_mult(alpha, self.handle, xv, yv)
            else:
                rc = L.spx_matvec_kernel(alpha, self.handle, xv, beta, yv)
        finally:
            L.spx_vec_destroy(xv)
            L.spx_vec_destroy(yv)
        if rc != SPX_SUCCESS:
            raise SpxError("spx_matvec failed (see stderr)")
        return y

    def partition(self):
        L = lib()
        p = L.spx_mat_get_partition(self.handle)
        n = self.info().nr_partitions
        rs = [L.spx_partition_get_rs(p)[i] for i in range(n)]
        re = [L.spx_partition_get_re(p)[i] for i in range(n)]
        L.spx_partition_destroy(p)
        return rs, re

    # -- device-resident extension (include/sparsex_hip.h) -------------------
    def hip_matvec_mult(self, alpha, x_ptr, y_ptr, stream=0):
        rc = lib().spx_hip_matvec_mult(alpha, self.handle, x_ptr, y_ptr, stream)
        if rc != SPX_SUCCESS:
            raise SpxError("spx_hip_matvec_mult failed (see stderr)")

    def hip_matvec_kernel(self, alpha, x_ptr, beta, y_ptr, stream=0):
        rc = lib().spx_hip_matvec_kernel(alpha, self.handle, x_ptr, beta, y_ptr, stream)
        if rc != SPX_SUCCESS:
            raise SpxError("spx_hip_matvec_kernel failed (see stderr)")

    def info(self):
        inf = HipInfo()
        if lib().spx_hip_mat_info(self.handle, C.byref(inf)) != SPX_SUCCESS:
            raise SpxError("spx_hip_mat_info failed")
        return inf

    def unit_windows(self, budget=4096, gap=16):
        """The unit windows of x planned from the stream (spx_hip_mat_unit_windows) as numpy copies:
        tab [n_rowblocks, 16, 2] u32, xdescs [n_descs, 2] u32, passes as raw bytes [n_passes, 24]."""
        pl = XwPlan()
        L = lib()
        L.spx_hip_mat_unit_windows.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(XwPlan)]
        if L.spx_hip_mat_unit_windows(self.handle, budget, gap, C.byref(pl)) != SPX_SUCCESS:
            raise SpxError("spx_hip_mat_unit_windows failed")
        out = {k: getattr(pl, k) for k in ("n_rowblocks", "n_descs", "n_passes", "rowblocks_with_windows",
                                           "rowblocks_with_units", "staged_doubles", "unit_elems",
                                           "unit_elems_lds", "lds_doubles")}
        out["tab"] = np.ctypeslib.as_array(pl.tab, shape=(pl.n_rowblocks * 32,)).reshape(-1, 16, 2).copy() \
            if pl.n_rowblocks else np.zeros((0, 16, 2), np.uint32)
        out["xdescs"] = np.ctypeslib.as_array(pl.xdescs, shape=(pl.n_descs * 2,)).reshape(-1, 2).copy() \
            if pl.n_descs else np.zeros((0, 2), np.uint32)
        raw = C.string_at(pl.passes, pl.n_passes * 24) if pl.n_passes else b""
        out["passes"] = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 24).copy()
        return out

    def sym_pipeline(self):
        """The device-side pass headers of the pipelined read-once kernel (spx_hip_mat_sym_pipeline) as numpy
        copies: (passes as a [n, 6] uint32 array, SX passes per row-block, the plan's counters)."""
        import numpy as np
        pl = SxPlan()
        L = lib()
        L.spx_hip_mat_sym_pipeline.argtypes = [C.c_void_p, C.POINTER(SxPlan)]
        if L.spx_hip_mat_sym_pipeline(self.handle, C.byref(pl)) != SPX_SUCCESS:
            raise SpxError("spx_hip_mat_sym_pipeline failed")
        passes = np.ctypeslib.as_array(C.cast(pl.passes, C.POINTER(C.c_uint32)), shape=(pl.n_passes, 6)).copy() \
            if pl.n_passes else np.zeros((0, 6), np.uint32)
        n_sx = np.ctypeslib.as_array(pl.n_sx, shape=(pl.n_rowblocks,)).copy() if pl.n_rowblocks else np.zeros(0, np.uint32)
        return passes, n_sx, {k: int(getattr(pl, k)) for k in ("rowblocks_with_sx", "sym_elems", "sx_elems",
                                                                "sym_passes", "sx_passes")}

    def host_parts(self):
        """Parts the last spx_matvec_* on host vectors ran in (spx_hip_mat_host_parts; 0: in one piece)."""
        L = lib()
        L.spx_hip_mat_host_parts.restype = C.c_int
        return int(L.spx_hip_mat_host_parts(C.c_void_p(self.handle)))

    def host_order(self):
        """The order those parts ran in where x went up piece by piece as they needed it (spx_hip_mat_host_order);
        [] when x went up whole or stayed resident."""
        L = lib()
        L.spx_hip_mat_host_order.restype = C.c_int
        L.spx_hip_mat_host_order.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int]
        buf = (C.c_int32 * 64)()
        k = L.spx_hip_mat_host_order(C.c_void_p(self.handle), buf, 64)
        return [int(buf[i]) for i in range(min(k, 64))]

    def x_pieces(self, piece):
        """Which pieces of x (of `piece` elements, at most 64 of them) every row-block reads (spx_hip_mat_x_pieces):
        (masks as uint64, first rows, row counts), numpy copies; host side."""
        import numpy as np
        L = lib()
        L.spx_hip_mat_x_pieces.restype = C.c_int64
        L.spx_hip_mat_x_pieces.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        n = L.spx_hip_mat_x_pieces(C.c_void_p(self.handle), piece, None, None, None, 0)
        if n < 0:
            raise SpxError("spx_hip_mat_x_pieces failed")
        m, r0, nr = np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        if n and L.spx_hip_mat_x_pieces(C.c_void_p(self.handle), piece, m.ctypes.data, r0.ctypes.data, nr.ctypes.data, n) != n:
            raise SpxError("spx_hip_mat_x_pieces failed")
        return m, r0, nr

    def export_csx(self, part=0):
        """Reference-format CSX arrays of one partition as numpy copies."""
        ex = CsxExport()
        if lib().spx_hip_mat_export_csx(self.handle, part, C.byref(ex)) != SPX_SUCCESS:
            raise SpxError("spx_hip_mat_export_csx failed")
        out = {
            "values": np.ctypeslib.as_array(ex.values, shape=(max(ex.nnz, 0),)).copy()
            if ex.nnz else np.zeros(0),
            "ctl": np.ctypeslib.as_array(ex.ctl, shape=(ex.ctl_size,)).copy()
            if ex.ctl_size else np.zeros(0, dtype=np.uint8),
            "nnz": ex.nnz, "ncols": ex.ncols, "nrows": ex.nrows,
            "row_start": ex.row_start, "row_jumps": int(ex.row_jumps),
            "full_colind": int(ex.full_colind),
            "id_map": [int(v) for v in ex.id_map],
            "rows_info": np.ctypeslib.as_array(ex.rows_info, shape=(ex.nrows, 3)).copy()
            if ex.nrows else np.zeros((0, 3), dtype=np.int32),
            "dvalues": (np.ctypeslib.as_array(ex.dvalues, shape=(ex.nrows,)).copy()
                        if bool(ex.dvalues) and ex.nrows else None),
        }
        return out

    def export_units(self, part=0):
        L = lib()
        n = L.spx_hip_mat_export_units(self.handle, part, None, 0)
        if n < 0:
            raise SpxError("spx_hip_mat_export_units failed")
        recs = (UnitRecord * max(n, 1))()
        L.spx_hip_mat_export_units(self.handle, part, recs, n)
        return [(r.type, r.delta, r.size, r.row, r.col) for r in recs[:n]]

    def get_entry(self, row, col, indexing=SPX_INDEX_ZERO_BASED):
        """``spx_mat_get_entry`` -- value of a stored nonzero (raises if absent)."""
        v = C.c_double(0.0)
        rc = lib().spx_mat_get_entry(C.c_void_p(self.handle), C.c_int(row), C.c_int(col),
                                     C.byref(v), C.c_int(indexing))
        if rc != SPX_SUCCESS:
            raise SpxError("entry (%d, %d) not found" % (row, col))
        return v.value

    def set_entry(self, row, col, value, indexing=SPX_INDEX_ZERO_BASED):
        """``spx_mat_set_entry`` -- overwrite the value of a stored nonzero."""
        rc = lib().spx_mat_set_entry(C.c_void_p(self.handle), C.c_int(row), C.c_int(col),
                                     C.c_double(value), C.c_int(indexing))
        if rc != SPX_SUCCESS:
            raise SpxError("entry (%d, %d) not set" % (row, col))

    def get_perm(self):
        """``spx_mat_get_perm`` -- perm[old index] = new index of a matrix tuned with
        ``reorder=True``; ``None`` when the matrix was not reordered."""
        L = lib()
        L.spx_mat_get_perm.restype = C.POINTER(C.c_int)
        L.spx_mat_get_perm.argtypes = [C.c_void_p]
        p = L.spx_mat_get_perm(self.handle)
        if not p:
            return None
        return np.ctypeslib.as_array(p, shape=(self.nrows,)).copy()

    def save(self, filename):
        """``spx_mat_save`` -- the tuned matrix (descriptor stream) to a file."""
        L = lib()
        L.spx_mat_save.argtypes = [C.c_void_p, C.c_char_p]
        if L.spx_mat_save(self.handle, str(filename).encode()) != SPX_SUCCESS:
            raise SpxError("spx_mat_save failed (see stderr)")

    def tune_log(self):
        return lib().spx_hip_mat_tune_log(self.handle).decode()

    def destroy(self):
        if self.handle:
            lib().spx_mat_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class DeviceVector:
    """``spx_hip_vec_t``: a vector resident in HBM with the BLAS-1 helpers of
    include/sparsex_hip.h (device counterparts of the reference's spx_vec_*)."""

    def __init__(self, size=None, host=None):
        L = lib()
        if host is not None:
            size = host.size
        self.handle = L.spx_hip_vec_create(size)
        if not self.handle:
            raise SpxError("spx_hip_vec_create failed (see stderr)")
        self.size = size
        if host is not None:
            self.upload(host)

    def _check(self, rc, what):
        if rc != SPX_SUCCESS:
            raise SpxError(what + " failed (see stderr)")

    def _view(self, arr):
        assert arr.dtype == np.float64 and arr.flags.c_contiguous
        return lib().spx_vec_create_from_buff(arr.ctypes.data_as(C.POINTER(C.c_double)), None,
                                              arr.size, None, SPX_VEC_AS_IS)

    def upload(self, arr, stream=0):
        v = self._view(arr)
        try:
            self._check(lib().spx_hip_vec_upload(self.handle, v, stream), "spx_hip_vec_upload")
        finally:
            lib().spx_vec_destroy(v)

    def download(self, stream=0):
        out = np.empty(self.size)
        v = self._view(out)
        try:
            self._check(lib().spx_hip_vec_download(self.handle, v, stream), "spx_hip_vec_download")
        finally:
            lib().spx_vec_destroy(v)
        return out

    def data_ptr(self):
        return lib().spx_hip_vec_data(self.handle)

    def init(self, val, stream=0):
        self._check(lib().spx_hip_vec_init(self.handle, val, stream), "spx_hip_vec_init")

    def scale_into(self, dst, num, stream=0):          # dst <- num * self
        self._check(lib().spx_hip_vec_scale(self.handle, dst.handle, num, stream), "spx_hip_vec_scale")

    def scale_add_into(self, other, dst, num, stream=0):   # dst <- self + num * other
        self._check(lib().spx_hip_vec_scale_add(self.handle, other.handle, dst.handle, num, stream),
                    "spx_hip_vec_scale_add")

    def dot(self, other, stream=0):
        r = C.c_double(0.0)
        self._check(lib().spx_hip_vec_mul(self.handle, other.handle, C.byref(r), stream),
                    "spx_hip_vec_mul")
        return r.value

    def probe_read_write(self, dst, chunk_doubles, write_doubles, stream=0):
        """Diagnostic (spx_hip_probe_read_write): this vector read in chunks, `write_doubles` stored per chunk to dst."""
        L = lib()
        L.spx_hip_probe_read_write.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
        self._check(L.spx_hip_probe_read_write(self.handle, dst.handle, chunk_doubles, write_doubles, C.c_void_p(stream)),
                    "spx_hip_probe_read_write")

    def copy_into(self, dst, stream=0):
        self._check(lib().spx_hip_vec_copy(self.handle, dst.handle, stream), "spx_hip_vec_copy")

    def destroy(self):
        if self.handle:
            lib().spx_hip_vec_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def matvec_kernel_vec(A, alpha, x, beta, y, stream=0):
    """``spx_hip_matvec_kernel_vec``: y <- alpha*A*x + beta*y on DeviceVectors."""
    rc = lib().spx_hip_matvec_kernel_vec(alpha, A.handle, x.handle, beta, y.handle, stream)
    if rc != SPX_SUCCESS:
        raise SpxError("spx_hip_matvec_kernel_vec failed (see stderr)")


def mat_restore(filename):
    """``spx_mat_restore`` -- a matrix saved by :meth:`Matrix.save`; no re-tuning."""
    L = lib()
    L.spx_mat_restore.restype = C.c_void_p
    L.spx_mat_restore.argtypes = [C.c_char_p]
    h = L.spx_mat_restore(str(filename).encode())
    if not h:
        raise SpxError("spx_mat_restore failed (see stderr)")
    return Matrix(h)


def _vec_permute(fn, v, perm):
    L = lib()
    assert v.dtype == np.float64 and v.flags.c_contiguous
    perm = np.ascontiguousarray(perm, dtype=np.int32)
    getattr(L, fn).argtypes = [C.POINTER(VectorStruct), C.POINTER(C.c_int)]
    vv = L.spx_vec_create_from_buff(v.ctypes.data_as(C.POINTER(C.c_double)), None,
                                    v.size, None, SPX_VEC_AS_IS)
    try:
        if getattr(L, fn)(vv, perm.ctypes.data_as(C.POINTER(C.c_int))) != SPX_SUCCESS:
            raise SpxError("%s failed" % fn)
    finally:
        L.spx_vec_destroy(vv)
    return v


def vec_reorder(v, perm):
    """``spx_vec_reorder(v, p)`` in place: new[p[i]] = old[i]."""
    return _vec_permute("spx_vec_reorder", v, perm)


def vec_inv_reorder(v, perm):
    """``spx_vec_inv_reorder(v, p)`` in place: new[i] = old[p[i]]."""
    return _vec_permute("spx_vec_inv_reorder", v, perm)


def matvec_kernel_csr(A, rowptr, colind, values, nrows, ncols, alpha, x, beta, y):
    """``spx_matvec_kernel_csr(&A, nrows, ncols, rowptr, colind, values, alpha, x, beta, y)``
    (reference src/api/matvec.c:622-673): with ``A`` None the CSR arrays (zero-based) are
    tuned first and the new :class:`Matrix` is returned; later calls pass it back and
    only multiply."""
    L = lib()
    rp = np.ascontiguousarray(rowptr, dtype=np.int32) if rowptr is not None else None
    ci = np.ascontiguousarray(colind, dtype=np.int32) if colind is not None else None
    va = np.ascontiguousarray(values, dtype=np.float64) if values is not None else None
    h = C.c_void_p(A.handle if A is not None else None)
    L.spx_matvec_kernel_csr.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_double, C.POINTER(VectorStruct), C.c_double,
                                        C.POINTER(VectorStruct)]

    def view(v):
        if v is None:
            return None
        return L.spx_vec_create_from_buff(v.ctypes.data_as(C.POINTER(C.c_double)), None, v.size, None,
                                          SPX_VEC_AS_IS)

    def ptr(a):
        return a.ctypes.data_as(C.c_void_p) if a is not None else None
    xv, yv = view(x), view(y)
    try:
        rc = L.spx_matvec_kernel_csr(C.byref(h), nrows, ncols, ptr(rp), ptr(ci), ptr(va), alpha, xv, beta, yv)
    finally:
        for v in (xv, yv):
            if v:
                L.spx_vec_destroy(v)
    if rc != SPX_SUCCESS:
        raise SpxError("spx_matvec_kernel_csr failed (see stderr)")
    if A is not None:
        return A
    M = Matrix(h.value)
    M._keep = (rp, ci, va)
    return M


def mat_tune(inp, reorder=False):
    """``spx_mat_tune(input[, SPX_MAT_REORDER])``."""
    h = lib().spx_mat_tune(C.c_void_p(inp.handle), C.c_int(SPX_MAT_REORDER if reorder else 0))
    if not h:
        raise SpxError("spx_mat_tune failed (see stderr)")
    return Matrix(h)


# ---- one process per GPU (include/sparsex_hip.h, "row-partitioned matrices") -----------

_XCHG_HOST = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_size_t),
                         C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.POINTER(C.c_size_t),
                         C.POINTER(C.c_size_t))
_XCHG_DEV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t),
                        C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t),
                        C.POINTER(C.c_size_t), C.c_void_p)


class TransportStruct(C.Structure):
    """``spx_hip_transport_t``."""
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_int), ("world", C.c_int),
                ("exchange_host", _XCHG_HOST), ("exchange_device", _XCHG_DEV)]


class DistPlanStruct(C.Structure):
    """``spx_hip_dist_plan_t``."""
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32),
                ("row_lo", C.POINTER(C.c_int)), ("row_hi", C.POINTER(C.c_int)),
                ("n_send", C.c_int64), ("send_rows", C.POINTER(C.c_int)),
                ("send_off", C.POINTER(C.c_size_t)), ("send_cnt", C.POINTER(C.c_size_t)),
                ("n_recv", C.c_int64),
                ("recv_off", C.POINTER(C.c_size_t)), ("recv_cnt", C.POINTER(C.c_size_t)),
                ("n_fix_rows", C.c_int64), ("fix_rows", C.POINTER(C.c_int)),
                ("fix_ptr", C.POINTER(C.c_uint32)), ("fix_pos", C.POINTER(C.c_uint32)),
                ("any_exchange", C.c_int32)]


class DistHaloStruct(C.Structure):
    """``spx_hip_dist_halo_t``."""
    _fields_ = [("n_recv", C.c_int64), ("recv_cols", C.POINTER(C.c_int)),
                ("recv_off", C.POINTER(C.c_size_t)), ("recv_cnt", C.POINTER(C.c_size_t)),
                ("n_send", C.c_int64), ("send_rows", C.POINTER(C.c_int)),
                ("send_off", C.POINTER(C.c_size_t)), ("send_cnt", C.POINTER(C.c_size_t))]


SPX_DIST_OWNED_ROWS = 0
SPX_DIST_GATHER_Y = 1
SPX_DIST_HALO_X = 2
SPX_DIST_OVERLAP = 4
SPX_RCCL_ID_BYTES = 128


def rccl_unique_id():
    """``spx_hip_rccl_unique_id``: 128 bytes for rank 0 to hand to the others."""
    buf = C.create_string_buffer(SPX_RCCL_ID_BYTES)
    if lib().spx_hip_rccl_unique_id(buf) != SPX_SUCCESS:
        raise SpxError("spx_hip_rccl_unique_id failed (see stderr)")
    return buf.raw


class RcclTransport:
    """The built-in transport: RCCL point-to-point over xGMI (collective to create)."""

    def __init__(self, unique_id, rank, world):
        L = lib()
        L.spx_hip_transport_rccl.restype = C.POINTER(TransportStruct)
        L.spx_hip_transport_rccl.argtypes = [C.c_char_p, C.c_int, C.c_int]
        L.spx_hip_transport_destroy.argtypes = [C.POINTER(TransportStruct)]
        L.spx_hip_transport_destroy.restype = None
        self.ptr = L.spx_hip_transport_rccl(unique_id, rank, world)
        if not self.ptr:
            raise SpxError("spx_hip_transport_rccl failed (see stderr)")
        self.rank, self.world = rank, world

    def rccl_ranks(self):
        """Ranks the RCCL communicator holds (ncclCommCount), -1 where librccl does not say."""
        L = lib()
        L.spx_hip_transport_rccl_ranks.argtypes = [C.POINTER(TransportStruct)]
        L.spx_hip_transport_rccl_ranks.restype = C.c_int
        return int(L.spx_hip_transport_rccl_ranks(self.ptr))

    def destroy(self):
        if self.ptr:
            lib().spx_hip_transport_destroy(self.ptr)
            self.ptr = None


class CallbackTransport:
    """A transport made of two Python callables (tests: torch.distributed/gloo).

    ``host(send, send_off, send_cnt, recv, recv_off, recv_cnt)`` gets numpy uint64 views;
    ``device(send_ptr, send_off, send_cnt, recv_ptr, recv_off, recv_cnt, stream)`` gets raw
    device addresses.  Offsets/counts are numpy arrays of length ``world``."""

    def __init__(self, rank, world, host, device):
        self.rank, self.world = rank, world

        def _arr(p):
            return np.ctypeslib.as_array(p, shape=(world,)).astype(np.int64)

        def _host(ctx, send, soff, scnt, recv, roff, rcnt):
            try:
                so, sc, ro, rc = _arr(soff), _arr(scnt), _arr(roff), _arr(rcnt)
                ns = int(max([so[q] + sc[q] for q in range(world) if q != rank] + [1]))
                nr = int(max([ro[q] + rc[q] for q in range(world) if q != rank] + [1]))
                host(np.ctypeslib.as_array(send, shape=(ns,)), so, sc,
                     np.ctypeslib.as_array(recv, shape=(nr,)), ro, rc)
                return 0
            except Exception as e:       # never unwind through C
                print("transport host exchange failed: %r" % (e,))
                return -1

        def _dev(ctx, send, soff, scnt, recv, roff, rcnt, stream):
            try:
                device(send or 0, _arr(soff), _arr(scnt), recv or 0, _arr(roff), _arr(rcnt), stream or 0)
                return 0
            except Exception as e:
                print("transport device exchange failed: %r" % (e,))
                return -1

        self._cb = (_XCHG_HOST(_host), _XCHG_DEV(_dev))      # keep the thunks alive
        self.struct = TransportStruct(None, rank, world, self._cb[0], self._cb[1])
        self.ptr = C.pointer(self.struct)

    def destroy(self):
        pass


SPX_DIST_REORDER_RCM = 1
SPX_DIST_REORDER_RCM_OWNER = 2
SPX_DIST_PATTERN_SYMMETRIC = 1


def dist_reorder(rowptr, colind, nrows, world, mode=SPX_DIST_REORDER_RCM_OWNER, pattern_symmetric=False):
    """``spx_hip_dist_reorder``: perm[old] = new (int32) for the zero-based CSR pattern of a square
    matrix that is to be dealt to ``world`` processes by nonzeros."""
    L = lib()
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    colind = np.ascontiguousarray(colind, dtype=np.int32)
    perm = np.empty(int(nrows), dtype=np.int32)
    L.spx_hip_dist_reorder.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    rc = L.spx_hip_dist_reorder(rowptr.ctypes.data, colind.ctypes.data, int(nrows), SPX_INDEX_ZERO_BASED, int(world),
                                int(mode), SPX_DIST_PATTERN_SYMMETRIC if pattern_symmetric else 0, perm.ctypes.data)
    if rc != SPX_SUCCESS:
        raise SpxError("spx_hip_dist_reorder failed (see stderr)")
    return perm


def _dist_attach(self, transport):
    """``spx_hip_mat_dist_attach`` (collective)."""
    L = lib()
    L.spx_hip_mat_dist_attach.argtypes = [C.c_void_p, C.POINTER(TransportStruct)]
    if L.spx_hip_mat_dist_attach(self.handle, transport.ptr) != SPX_SUCCESS:
        raise SpxError("spx_hip_mat_dist_attach failed (see stderr)")
    self._transport = transport


def _dist_plan(self):
    """``spx_hip_mat_dist_plan`` as a dict of numpy copies."""
    L = lib()
    L.spx_hip_mat_dist_plan.argtypes = [C.c_void_p, C.POINTER(DistPlanStruct)]
    p = DistPlanStruct()
    if L.spx_hip_mat_dist_plan(self.handle, C.byref(p)) != SPX_SUCCESS:
        raise SpxError("spx_hip_mat_dist_plan failed (see stderr)")
    W = p.world

    def arr(ptr, n, dt):
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt) if n else np.zeros(0, dtype=dt)
    return {"rank": p.rank, "world": W, "row_lo": arr(p.row_lo, W, np.int64), "row_hi": arr(p.row_hi, W, np.int64),
            "send_rows": arr(p.send_rows, p.n_send, np.int64),
            "send_off": arr(p.send_off, W, np.int64), "send_cnt": arr(p.send_cnt, W, np.int64),
            "n_recv": int(p.n_recv),
            "recv_off": arr(p.recv_off, W, np.int64), "recv_cnt": arr(p.recv_cnt, W, np.int64),
            "fix_rows": arr(p.fix_rows, p.n_fix_rows, np.int64),
            "fix_ptr": arr(p.fix_ptr, p.n_fix_rows + 1 if p.n_fix_rows else 1, np.int64),
            "fix_pos": arr(p.fix_pos, p.n_recv, np.int64), "any_exchange": bool(p.any_exchange)}


def _dist_halo(self):
    """``spx_hip_mat_dist_halo`` as a dict of numpy copies: the entries of x this process needs
    of the others (``recv_cols``, by owner) and the own entries the others asked for."""
    L = lib()
    L.spx_hip_mat_dist_halo.argtypes = [C.c_void_p, C.POINTER(DistHaloStruct)]
    h = DistHaloStruct()
    if L.spx_hip_mat_dist_halo(self.handle, C.byref(h)) != SPX_SUCCESS:
        raise SpxError("spx_hip_mat_dist_halo failed (see stderr)")
    W = int(self.dist_plan()["world"])

    def arr(ptr, n):
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(np.int64) if n else np.zeros(0, dtype=np.int64)
    return {"recv_cols": arr(h.recv_cols, h.n_recv), "recv_off": arr(h.recv_off, W), "recv_cnt": arr(h.recv_cnt, W),
            "send_rows": arr(h.send_rows, h.n_send), "send_off": arr(h.send_off, W), "send_cnt": arr(h.send_cnt, W)}


def _hip_matvec_parts(self, alpha, x_ptr, beta, y_ptr, parts, stream=0):
    """``spx_hip_matvec_parts``: the product in ``parts`` launches over consecutive parts of the rows."""
    L = lib()
    L.spx_hip_matvec_parts.argtypes = [C.c_double, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_void_p,
                                       C.POINTER(C.c_int)]
    n = C.c_int(0)
    if L.spx_hip_matvec_parts(alpha, self.handle, x_ptr, beta, y_ptr, int(parts), stream, C.byref(n)) != SPX_SUCCESS:
        raise SpxError("spx_hip_matvec_parts failed (see stderr)")
    return n.value


def _dist_parts(self):
    """``spx_hip_mat_dist_parts``: launches this process' product is cut into in the overlapped step."""
    L = lib()
    L.spx_hip_mat_dist_parts.argtypes = [C.c_void_p]
    return int(L.spx_hip_mat_dist_parts(self.handle))


def _dist_rounds(self):
    """The rounds of the overlapped step as a list of dicts of [world] arrays (send_off, send_cnt,
    recv_off, recv_cnt): segments of the halo lists that travel in every round."""
    L = lib()
    L.spx_hip_mat_dist_rounds.argtypes = [C.c_void_p]
    L.spx_hip_mat_dist_round.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.POINTER(C.c_size_t))] * 4
    W = int(self.dist_plan()["world"])
    out = []
    for r in range(L.spx_hip_mat_dist_rounds(self.handle)):
        ptrs = [C.POINTER(C.c_size_t)() for _ in range(4)]
        if L.spx_hip_mat_dist_round(self.handle, r, *[C.byref(p) for p in ptrs]) != SPX_SUCCESS:
            raise SpxError("spx_hip_mat_dist_round failed")
        out.append({k: np.ctypeslib.as_array(p, shape=(W,)).astype(np.int64)
                    for k, p in zip(("send_off", "send_cnt", "recv_off", "recv_cnt"), ptrs)})
    return out


def _hip_matvec_dist(self, alpha, x_ptr, beta, y_ptr, flags=SPX_DIST_OWNED_ROWS, stream=0):
    """``spx_hip_matvec_dist`` (collective)."""
    L = lib()
    L.spx_hip_matvec_dist.argtypes = [C.c_double, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p,
                                      C.c_int, C.c_void_p]
    if L.spx_hip_matvec_dist(alpha, self.handle, x_ptr, beta, y_ptr, flags, stream) != SPX_SUCCESS:
        raise SpxError("spx_hip_matvec_dist failed (see stderr)")


Matrix.dist_attach = _dist_attach
Matrix.dist_plan = _dist_plan
Matrix.dist_halo = _dist_halo
Matrix.dist_rounds = _dist_rounds
Matrix.dist_parts = _dist_parts
Matrix.hip_matvec_parts = _hip_matvec_parts
Matrix.hip_matvec_dist = _hip_matvec_dist
