"""MI355X-native CSX SpMV behind the SparseX C API.

The product is ``lib/libsparsex.so`` (host CSX preprocessor in C++ + HIP
interpreter kernel for gfx950).  This package is the thin host-side mirror of
the reference's C interface (``include/sparsex/matvec.h`` in the reference
tree): same function names, argument meaning and error behaviour, bound with
ctypes, plus helpers to hand HBM-resident torch tensors to the library.
"""
from .api import (  # noqa: F401
    SpxError, lib, lib_path, Matrix, Input, option_set, options_reset,
    input_load_csr, input_load_mmf, mat_tune, mat_restore, DeviceVector, matvec_kernel_vec, matvec_kernel_csr,
    vec_reorder, vec_inv_reorder, RcclTransport, CallbackTransport, rccl_unique_id,
    SPX_DIST_OWNED_ROWS, SPX_DIST_GATHER_Y, SPX_DIST_HALO_X, SPX_DIST_OVERLAP, dist_reorder,
    SPX_DIST_REORDER_RCM, SPX_DIST_REORDER_RCM_OWNER,
    SPX_SUCCESS, SPX_FAILURE, SPX_INDEX_ZERO_BASED, SPX_INDEX_ONE_BASED,
    SPX_MAT_REORDER, SPX_VEC_AS_IS, SPX_VEC_TUNE,
)
