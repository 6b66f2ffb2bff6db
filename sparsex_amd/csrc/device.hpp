// device.hpp -- HBM-resident form of a tuned matrix and the launch entry
// points of the HIP interpreter kernel (implemented in spmv_kernels.hip).
//
// Replaces the reference's JIT'd per-partition spmv_fn + thread-pool dispatch
// (src/internals/CsxKernels.cpp:35-129, src/internals/CsxSpmv.cpp:28-86).
#pragma once

#include "gpu_emit.hpp"

#include <functional>
#include <string>
#include <vector>

namespace spx {

struct DeviceMatrix;   // opaque; owns device allocations

// Throws FatalError (message includes the HIP error string) on any failure.
int device_count();                       // 0 when no usable HIP device
DeviceMatrix *device_upload(const GpuStream &s, size_t nrows, size_t ncols,
                            bool symmetric, idx_t own_row_lo, idx_t own_row_hi,
                            int device);
void device_free(DeviceMatrix *m);

// y <- alpha*A*x + beta*y on device pointers, asynchronous on `stream`
// (a hipStream_t passed as void*; NULL = default stream).
void device_spmv(DeviceMatrix *m, double alpha, const double *d_x, double beta,
                 double *d_y, void *stream);

// The product in K launches over consecutive parts of the row-blocks (equal work each), so that
// the exchange of a row-partitioned matrix can start on the rows of part k while part k + 1 is
// computed (dist.cpp).  Plain general streams only: device_plan_chunks returns 0 for the others;
// `row_bounds` receives the first row of every part and the end of the own rows.
// `slot`: 0 = the cut of an attached exchange plan, 1 = a caller's own (spx_hip_matvec_parts): independent of each other
size_t device_plan_chunks(DeviceMatrix *m, size_t K, std::vector<size_t> &row_bounds, int slot = 0);
// `position` (symmetric streams, whose parts may run in any order): bit 0 = this part is launched first, bit 1 = last;
// -1 = by its number
void device_spmv_chunk(DeviceMatrix *m, size_t k, double alpha, const double *d_x, double beta,
                       double *d_y, void *stream, int slot = 0, int position = -1);

// host-vector convenience path used by spx_matvec_*: H2D x (and y when
// beta != 0), kernel, D2H y; synchronous.  Vectors the library allocated itself
// are pinned (device_host_alloc) and are copied from / to directly; user
// buffers go through pinned staging copies.
// `after(d_y, stream)`, if given, runs between the kernel and the copy back (the
// exchange of a row-partitioned matrix).
// `x_version` != 0 names the contents of h_x (spx.vec.device: library-created
// vectors carry a version that every spx_vec_* mutator advances): the copy of x in
// HBM is reused while the version stays the same.
void device_spmv_host(DeviceMatrix *m, double alpha, const double *h_x, bool x_pinned,
                      double beta, double *h_y, bool y_pinned,
                      const std::function<void(double *, void *)> &after = nullptr,
                      uint64_t x_version = 0);

// true while `stream` (a hipStream_t) is being captured into a hipGraph
bool device_stream_is_capturing(void *stream);

// page-locked host memory for the library's own vectors; nullptr when there is
// no HIP device (the caller falls back to malloc)
void *device_host_alloc(size_t bytes);
void device_host_free(void *p);
// a client's buffer page-locked where it lies (1: done, 2: it already was, 0: not possible) / released again
size_t device_host_parts_min_bytes();    // vectors of this size and more: y back in parts (32 MB; tests lower it)
int device_host_register(void *p, size_t bytes);
void device_host_unregister(void *p);

// symmetric slice: the first kernel clears y on rows [first_row, own rows) only
// (default 0: the whole partial vector is defined, for a caller-side all-reduce)
void device_set_init_rows(DeviceMatrix *m, size_t first_row);

// symmetric tiles: hand the transposed sums over with global atomics (true) or
// through the spill array and a second kernel (false)
void device_set_sym_atomic(DeviceMatrix *m, bool on);
bool device_get_sym_atomic(const DeviceMatrix *m);
bool device_has_spill(const DeviceMatrix *m);

// spx.gpu.deterministic: every wavefront of a workgroup adds into a y tile of its own, the
// copies are summed in wavefront order -- repeated products are bit-identical
void device_set_deterministic(DeviceMatrix *m, bool on);
bool device_get_deterministic(const DeviceMatrix *m);
// the per-wavefront tiles alone (also a speed option: no two wavefronts add to the same LDS
// address; spx_mat_tune measures it on matrices without symmetric tiles)
void device_set_wave_tiles(DeviceMatrix *m, bool on);
bool device_get_wave_tiles(const DeviceMatrix *m);
bool device_has_tiles(const DeviceMatrix *m);

// wavefronts per workgroup of the SpMV kernel: 2, 4 or 8
void device_set_waves(DeviceMatrix *m, int waves);
int device_get_waves(const DeviceMatrix *m);

// unit windows of x in LDS + pipelined unit passes (csx_spmv_xw_kernel; xwindows.hpp): available where
// the stream was uploaded with a window budget and some row-block's columns fit it
bool device_has_xw(const DeviceMatrix *m);
int device_host_order(const DeviceMatrix *m, int32_t *order, int cap);   // the order of those parts when x went up piece by piece as they needed it (returns their number; 0: x went up whole or not at all)
void device_set_host_parts(DeviceMatrix *m, size_t parts);   // spx.rt.host_parts: 0 = the built-in choice, else at most 64
int device_host_parts(const DeviceMatrix *m);   // parts of the last product on host vectors whose y went back part by part (0: whole)
void device_set_xw(DeviceMatrix *m, bool on);
bool device_get_xw(const DeviceMatrix *m);
void device_xw_info(const DeviceMatrix *m, uint64_t &elems_lds, uint64_t &unit_elems, uint64_t &staged_doubles,
                    uint32_t &lds_bytes);

// the read-once passes of a symmetric stream pipelined (csx_spmv_sx_kernel; sxplan.hpp): available where the
// stream holds read-once row segments, no tiles, and was uploaded with GpuStream::sx_plan
bool device_has_sx(const DeviceMatrix *m);
void device_set_sx(DeviceMatrix *m, bool on);
bool device_get_sx(const DeviceMatrix *m);
void device_sx_info(const DeviceMatrix *m, uint64_t &elems_sx, uint64_t &elems_sym, size_t &rowblocks);

// seconds per SpMV (alpha = 1, beta = 0) over `launches` back-to-back launches
// on a private stream with scratch vectors -- what spx_mat_tune() measures to
// choose launch parameters
double device_time_spmv(DeviceMatrix *m, int warmup, int launches);

// copies the descriptor stream back from HBM (for spx_mat_save)
void device_download(const DeviceMatrix *m, GpuStream &s);

// one stored value (diagonal: of the symmetric path's diagonal array) read from /
// written to HBM where it lies (spx_mat_get_entry / spx_mat_set_entry);
// synchronous
double device_peek(const DeviceMatrix *m, bool diagonal, size_t index);
void device_poke(DeviceMatrix *m, bool diagonal, size_t index, double value);
void device_poke_mirror(DeviceMatrix *m, size_t index, double value);   // GpuStream::mirror_val

struct DeviceMatrixInfo {
    size_t n_rowblocks, n_shared_rows;
    size_t value_bytes, index_bytes;
    int device;
};
void device_info(const DeviceMatrix *m, DeviceMatrixInfo &info);

}  // namespace spx
