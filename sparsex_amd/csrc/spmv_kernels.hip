// spmv_kernels.hip -- the CSX interpreter for gfx950 (MI355X).
//
// One wavefront walks one row-block of the descriptor stream (gpu_format.h):
// coalesced 32-byte-per-lane reads of the packed values, segment-start bits
// ranked with ballot/mbcnt to find each nonzero's unit descriptor, strided
// decode of (row, col), gathered x, wave-level segmented reduction, an LDS
// y tile per wavefront, and one coalesced write of the owned rows of y.
//
// Semantics restated from the reference's SpMV templates
// (src/templates/csx_spmv_tmpl.c:66-101 and the per-unit bodies
// delta/horiz/vert/diag/rdiag/block_row/block_col _tmpl.c; symmetric:
// csx_sym_spmv_tmpl.c:60-106): every stored nonzero a(r,c) contributes
// alpha*a*x[c] to y[r] (and alpha*a*x[r] to y[c] on the symmetric path).
#include "device.hpp"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace spx {

#define HIP_CHECK(expr)                                                         \
    do {                                                                        \
        hipError_t e_ = (expr);                                                 \
        if (e_ != hipSuccess) {                                                 \
            std::string m_ = std::string("HIP failure: ") + #expr + ": " +      \
                             hipGetErrorString(e_);                             \
            log_msg(LOG_ERR, "%s\n", m_.c_str());                               \
            throw FatalError(m_);                                               \
        }                                                                       \
    } while (0)

struct KernelArgs {
    const SpxRowBlock *rbs;
    const double *values;
    const SpxUnitDesc *descs;
    const uint32_t *bits;
    const uint8_t *cidx;
    const uint16_t *segrows;
    const double *x;
    double *y;
    double *carry;
    double alpha, beta;
    uint32_t n_rb;
};

constexpr int WAVES_PER_BLOCK = 4;

__device__ __forceinline__ uint32_t lanes_below(uint64_t mask)
{
    // number of set bits of `mask` in lanes below the caller
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
}

// Adds every lane's runs of equal rows into the wavefront's LDS tile.
// rows[]/prods[] hold SPX_LANE_ELEMS consecutive nonzeros; n_valid of them
// (a prefix) are real.  Consecutive lanes hold consecutive nonzeros.
__device__ __forceinline__ void reduce_into_tile(double *tile, const int (&rows)[SPX_LANE_ELEMS],
                                                 const double (&prods)[SPX_LANE_ELEMS],
                                                 int n_valid, int lane)
{
    const bool active = n_valid > 0;
    int cur_row = active ? rows[0] : -1;
    double acc = active ? prods[0] : 0.0;
    bool multi = false;
#pragma unroll
    for (int j = 1; j < SPX_LANE_ELEMS; ++j) {
        if (j < n_valid) {
            if (rows[j] == cur_row) {
                acc += prods[j];
            } else {
                atomicAdd(&tile[cur_row], acc);   // a run that ends inside the lane
                cur_row = rows[j];
                acc = prods[j];
                multi = true;
            }
        }
    }
    // the lane's last run may continue in the next lanes: segmented scan
    const int prev_row = __shfl_up(cur_row, 1);
    int head = (lane == 0) || multi || !active || (prev_row != cur_row);
    const int head0 = head;
    if (!__all(head0)) {
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double a2 = __shfl_up(acc, d);
            const int h2 = __shfl_up(head, d);
            if (lane >= d && !head) {
                acc += a2;
                head |= h2;
            }
        }
    }
    const int next_head = __shfl_down(head0, 1);
    if (active && (lane == 63 || next_head)) atomicAdd(&tile[cur_row], acc);
}

template <bool SYM>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK)
void csx_spmv_kernel(KernelArgs a)
{
    __shared__ double tiles[WAVES_PER_BLOCK][SPX_MAX_RB_ROWS];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // XCD-aware order: workgroup b runs on XCD b % 8; give each XCD one
    // contiguous eighth of the row-blocks (gridDim.x is a multiple of 8)
    const uint32_t nb = gridDim.x;
    const uint32_t lb = (blockIdx.x & 7u) * (nb >> 3) + (blockIdx.x >> 3);
    const uint32_t rb_idx =
        __builtin_amdgcn_readfirstlane(lb * WAVES_PER_BLOCK + (uint32_t) wave);
    if (rb_idx >= a.n_rb) return;

    const SpxRowBlock rb = a.rbs[rb_idx];
    double *tile = tiles[wave];
    const int n_rows = rb.n_rows;
    for (int i = lane; i < n_rows; i += 64) tile[i] = 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

    const double *x = a.x;

    // ---------------- unit region ------------------------------------------------
    {
        const int n = rb.n_unit_elems;
        const double *vals = a.values + rb.val_off;
        const uint32_t *bits = a.bits + rb.bits_off;
        const SpxUnitDesc *descs = a.descs + rb.desc_off;
        uint32_t rank_base = 0;
        for (int base = 0; base < n; base += SPX_PASS_ELEMS) {
            const int e0 = base + lane * SPX_LANE_ELEMS;
            const uint32_t w = bits[(base >> 5) + (lane >> 3)];
            const uint32_t nib = (w >> ((lane & 7) * 4)) & 0xFu;
            const uint64_t m0 = __ballot(nib & 1u), m1 = __ballot(nib & 2u),
                           m2 = __ballot(nib & 4u), m3 = __ballot(nib & 8u);
            uint32_t rank = rank_base + lanes_below(m0) + lanes_below(m1) +
                            lanes_below(m2) + lanes_below(m3);
            rank_base += __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);

            int rows[SPX_LANE_ELEMS];
            double prods[SPX_LANE_ELEMS];
            const int n_valid = min(max(n - e0, 0), SPX_LANE_ELEMS);
            if (n_valid > 0) {
                const double2 v01 = *reinterpret_cast<const double2 *>(vals + e0);
                const double2 v23 = *reinterpret_cast<const double2 *>(vals + e0 + 2);
                const double v[SPX_LANE_ELEMS] = {v01.x, v01.y, v23.x, v23.y};
                SpxUnitDesc d;
#pragma unroll
                for (int j = 0; j < SPX_LANE_ELEMS; ++j) {
                    const bool starts = (nib >> j) & 1u;
                    rank += starts;
                    if (j == 0 || starts) d = descs[rank - 1];
                    const int k = e0 + j - (int) d.estart;
                    int in = 0, out = k;
                    if (d.mod) {
                        out = (int) (((float) k + 0.5f) * __frcp_rn((float) d.mod));
                        in = k - out * (int) d.mod;
                    }
                    const int r = (int) d.row0 + out * (int) d.drow_out + ((d.inner & 1) ? in : 0);
                    const int c = (int) d.col0 + out * d.dcol_out + ((d.inner & 1) ? 0 : in);
                    rows[j] = r;
                    if (j < n_valid) {
                        const double xv = x[c];
                        prods[j] = v[j] * xv;
                        if (SYM) atomicAdd(&a.y[c], a.alpha * v[j] * x[rb.row0 + r]);
                    } else {
                        prods[j] = 0.0;
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < SPX_LANE_ELEMS; ++j) { rows[j] = -1; prods[j] = 0.0; }
            }
            reduce_into_tile(tile, rows, prods, n_valid, lane);
        }
    }

    // ---------------- delta region ---------------------------------------------------
    {
        const int n = rb.n_delta_elems;
        const int n_unit_padded = (rb.n_unit_elems + 3) & ~3;
        const double *vals = a.values + rb.val_off + n_unit_padded;
        const uint32_t *bits =
            a.bits + rb.bits_off +
            ((rb.n_unit_elems + SPX_PASS_ELEMS - 1) / SPX_PASS_ELEMS) * SPX_PASS_WORDS;
        const uint16_t *segrows = a.segrows + rb.seg_off;
        const uint8_t *cidx = a.cidx + rb.cidx_off;
        const bool wide = rb.cidx_width == 4;
        uint32_t rank_base = 0;
        for (int base = 0; base < n; base += SPX_PASS_ELEMS) {
            const int e0 = base + lane * SPX_LANE_ELEMS;
            const uint32_t w = bits[(base >> 5) + (lane >> 3)];
            const uint32_t nib = (w >> ((lane & 7) * 4)) & 0xFu;
            const uint64_t m0 = __ballot(nib & 1u), m1 = __ballot(nib & 2u),
                           m2 = __ballot(nib & 4u), m3 = __ballot(nib & 8u);
            uint32_t rank = rank_base + lanes_below(m0) + lanes_below(m1) +
                            lanes_below(m2) + lanes_below(m3);
            rank_base += __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);

            int rows[SPX_LANE_ELEMS];
            double prods[SPX_LANE_ELEMS];
            const int n_valid = min(max(n - e0, 0), SPX_LANE_ELEMS);
            if (n_valid > 0) {
                const double2 v01 = *reinterpret_cast<const double2 *>(vals + e0);
                const double2 v23 = *reinterpret_cast<const double2 *>(vals + e0 + 2);
                const double v[SPX_LANE_ELEMS] = {v01.x, v01.y, v23.x, v23.y};
                uint32_t off[SPX_LANE_ELEMS];
                if (wide) {
                    const uint4 o = *reinterpret_cast<const uint4 *>(cidx + (size_t) e0 * 4);
                    off[0] = o.x; off[1] = o.y; off[2] = o.z; off[3] = o.w;
                } else {
                    const uint2 o = *reinterpret_cast<const uint2 *>(cidx + (size_t) e0 * 2);
                    off[0] = o.x & 0xffffu; off[1] = o.x >> 16;
                    off[2] = o.y & 0xffffu; off[3] = o.y >> 16;
                }
                int r = 0;
#pragma unroll
                for (int j = 0; j < SPX_LANE_ELEMS; ++j) {
                    const bool starts = (nib >> j) & 1u;
                    rank += starts;
                    if (j == 0 || starts) r = segrows[rank - 1];
                    rows[j] = r;
                    if (j < n_valid) {
                        const uint32_t c = rb.cbase + off[j];
                        prods[j] = v[j] * x[c];
                        if (SYM) atomicAdd(&a.y[c], a.alpha * v[j] * x[rb.row0 + r]);
                    } else {
                        prods[j] = 0.0;
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < SPX_LANE_ELEMS; ++j) { rows[j] = -1; prods[j] = 0.0; }
            }
            reduce_into_tile(tile, rows, prods, n_valid, lane);
        }
    }

    // ---------------- write the owned rows ------------------------------------------------
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (SYM) {
        for (int i = lane; i < n_rows; i += 64)
            atomicAdd(&a.y[rb.row0 + i], a.alpha * tile[i]);
    } else if (rb.flags & SPX_RB_SHARED) {
        if (lane == 0) a.carry[rb.carry_slot] = tile[0];
    } else if (a.beta == 0.0) {
        for (int i = lane; i < n_rows; i += 64) a.y[rb.row0 + i] = a.alpha * tile[i];
    } else {
        for (int i = lane; i < n_rows; i += 64) {
            const size_t g = (size_t) rb.row0 + i;
            a.y[g] = a.alpha * tile[i] + a.beta * a.y[g];
        }
    }
}

// rows split over several row-blocks: sum their partials
__global__ void csx_fixup_kernel(const SpxSharedRow *shared, uint32_t n_shared,
                                 const double *carry, double *y, double alpha,
                                 double beta)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_shared) return;
    const SpxSharedRow sr = shared[i];
    double s = 0.0;
    for (uint32_t k = 0; k < sr.n_slots; ++k) s += carry[sr.first_slot + k];
    y[sr.row] = (beta == 0.0) ? alpha * s : alpha * s + beta * y[sr.row];
}

// symmetric path, first step: y <- beta*y + alpha*diag(A)*x on the owned
// rows, 0 elsewhere (the main kernel then accumulates with atomics; on
// several GPUs the per-GPU vectors are summed afterwards)
__global__ void csx_sym_init_kernel(double *y, const double *x, const double *dvalues,
                                    size_t nrows, size_t own_lo, size_t own_hi,
                                    double alpha, double beta)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows) return;
    double v = 0.0;
    if (i >= own_lo && i < own_hi) {
        v = alpha * dvalues[i] * x[i];
        if (beta != 0.0) v += beta * y[i];
    }
    y[i] = v;
}

// ---- host side ------------------------------------------------------------------------------

struct DeviceMatrix {
    int device = 0;
    size_t nrows = 0, ncols = 0;
    bool symmetric = false;
    size_t own_lo = 0, own_hi = 0;
    uint32_t n_rb = 0, n_shared = 0, n_carry = 0;
    SpxRowBlock *rbs = nullptr;
    double *values = nullptr;
    SpxUnitDesc *descs = nullptr;
    uint32_t *bits = nullptr;
    uint8_t *cidx = nullptr;
    uint16_t *segrows = nullptr;
    SpxSharedRow *shared = nullptr;
    double *carry = nullptr;
    double *dvalues = nullptr;
    // staging vectors of the host-pointer path
    double *d_x = nullptr, *d_y = nullptr;
    size_t value_bytes = 0, index_bytes = 0;
};

int device_count()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

template <typename T>
static T *upload(const std::vector<T> &v, size_t slack_elems = 0)
{
    size_t bytes = (v.size() + slack_elems) * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    T *d = nullptr;
    HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&d), bytes));
    HIP_CHECK(hipMemset(d, 0, bytes));
    if (!v.empty())
        HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

DeviceMatrix *device_upload(const GpuStream &s, size_t nrows, size_t ncols,
                            bool symmetric, idx_t own_lo, idx_t own_hi, int device)
{
    if (device_count() <= 0) {
        log_msg(LOG_ERR, "no usable HIP device: the SpMV path of this library runs "
                "on an MI355X only (set spx.rt.host_only=true to tune without one)\n");
        throw FatalError("no HIP device");
    }
    if (device >= 0) HIP_CHECK(hipSetDevice(device));
    DeviceMatrix *m = new DeviceMatrix;
    HIP_CHECK(hipGetDevice(&m->device));
    m->nrows = nrows;
    m->ncols = ncols;
    m->symmetric = symmetric;
    m->own_lo = (size_t) own_lo;
    m->own_hi = (size_t) own_hi;
    m->n_rb = (uint32_t) s.rbs.size();
    m->n_shared = (uint32_t) s.shared.size();
    m->n_carry = s.n_carry;
    m->rbs = upload(s.rbs);
    m->values = upload(s.values, 8);
    m->descs = upload(s.descs, 1);
    m->bits = upload(s.bits, SPX_PASS_WORDS);
    m->cidx = upload(s.cidx, 64);
    m->segrows = upload(s.segrows, 8);
    m->shared = upload(s.shared);
    std::vector<double> zero_carry(s.n_carry ? s.n_carry : 1, 0.0);
    m->carry = upload(zero_carry);
    if (symmetric) {
        std::vector<double> dv = s.dvalues;
        dv.resize(nrows, 0.0);
        m->dvalues = upload(dv);
    }
    m->value_bytes = s.values.size() * sizeof(double);
    m->index_bytes = s.index_bytes();
    return m;
}

void device_free(DeviceMatrix *m)
{
    if (!m) return;
    (void) hipFree(m->rbs); (void) hipFree(m->values); (void) hipFree(m->descs);
    (void) hipFree(m->bits);
    (void) hipFree(m->cidx); (void) hipFree(m->segrows); (void) hipFree(m->shared);
    (void) hipFree(m->carry);
    if (m->dvalues) (void) hipFree(m->dvalues);
    if (m->d_x) (void) hipFree(m->d_x);
    if (m->d_y) (void) hipFree(m->d_y);
    delete m;
}

void device_spmv(DeviceMatrix *m, double alpha, const double *d_x, double beta,
                 double *d_y, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KernelArgs a;
    a.rbs = m->rbs; a.values = m->values; a.descs = m->descs; a.bits = m->bits;
    a.cidx = m->cidx; a.segrows = m->segrows; a.x = d_x; a.y = d_y;
    a.carry = m->carry; a.alpha = alpha; a.beta = beta; a.n_rb = m->n_rb;
    uint32_t blocks = (m->n_rb + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    blocks = (blocks + 7u) & ~7u;
    if (m->symmetric) {
        const int t = 256;
        hipLaunchKernelGGL(csx_sym_init_kernel, dim3((unsigned)((m->nrows + t - 1) / t)),
                           dim3(t), 0, stream, d_y, d_x, m->dvalues, m->nrows,
                           m->own_lo, m->own_hi, alpha, beta);
        if (blocks)
            hipLaunchKernelGGL(csx_spmv_kernel<true>, dim3(blocks),
                               dim3(64 * WAVES_PER_BLOCK), 0, stream, a);
    } else {
        if (blocks)
            hipLaunchKernelGGL(csx_spmv_kernel<false>, dim3(blocks),
                               dim3(64 * WAVES_PER_BLOCK), 0, stream, a);
        if (m->n_shared)
            hipLaunchKernelGGL(csx_fixup_kernel, dim3((m->n_shared + 63) / 64), dim3(64),
                               0, stream, m->shared, m->n_shared, m->carry, d_y, alpha,
                               beta);
    }
    HIP_CHECK(hipGetLastError());
}

void device_spmv_host(DeviceMatrix *m, double alpha, const double *h_x, double beta,
                      double *h_y)
{
    HIP_CHECK(hipSetDevice(m->device));
    if (!m->d_x) HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&m->d_x),
                                     (m->ncols ? m->ncols : 1) * sizeof(double)));
    if (!m->d_y) HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&m->d_y),
                                     (m->nrows ? m->nrows : 1) * sizeof(double)));
    HIP_CHECK(hipMemcpy(m->d_x, h_x, m->ncols * sizeof(double), hipMemcpyHostToDevice));
    // rows outside this process' slice keep the caller's values
    HIP_CHECK(hipMemcpy(m->d_y, h_y, m->nrows * sizeof(double), hipMemcpyHostToDevice));
    device_spmv(m, alpha, m->d_x, beta, m->d_y, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(h_y, m->d_y, m->nrows * sizeof(double), hipMemcpyDeviceToHost));
}

void device_info(const DeviceMatrix *m, DeviceMatrixInfo &info)
{
    info.n_rowblocks = m->n_rb;
    info.n_shared_rows = m->n_shared;
    info.value_bytes = m->value_bytes;
    info.index_bytes = m->index_bytes;
    info.device = m->device;
}

}  // namespace spx
