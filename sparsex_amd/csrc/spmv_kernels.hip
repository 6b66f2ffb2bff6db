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
    const uint16_t *pass_rank;
    const uint8_t *cidx;
    const uint16_t *segrows;
    const double *x;
    double *y;
    double *carry;
    double alpha, beta;
    uint32_t n_rb;
    uint32_t ablate;   // debugging only (env SPX_ABLATE): 1 no reduction, 2 no x gather
};

constexpr int WAVES_PER_BLOCK = 4;
constexpr int BLOCK_THREADS = 64 * WAVES_PER_BLOCK;

__device__ __forceinline__ uint32_t lanes_below(uint64_t mask)
{
    // number of set bits of `mask` in lanes below the caller
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
}

// Segment starts in front of the caller's first nonzero of this pass
// (nib = the lane's SPX_LANE_ELEMS start bits), not counting its own.
__device__ __forceinline__ uint32_t rank_before(uint32_t nib)
{
    const uint64_t m0 = __ballot(nib & 1u), m1 = __ballot(nib & 2u),
                   m2 = __ballot(nib & 4u), m3 = __ballot(nib & 8u);
    return lanes_below(m0) + lanes_below(m1) + lanes_below(m2) + lanes_below(m3);
}

// Adds the lane's SPX_LANE_ELEMS products into the row-block's LDS tile.
// Consecutive lanes hold consecutive nonzeros.  Products of equal rows are
// first merged inside the lane (branch-free run sums); when many lanes
// continue their neighbour's row (long rows) a segmented wave scan merges
// across lanes as well, otherwise every run end adds to the tile directly.
// valid[j] == false marks padding behind the region's last nonzero.
template <bool FULL>
__device__ __forceinline__ void reduce_into_tile(double *tile, const int (&rows)[SPX_LANE_ELEMS],
                                                 const double (&prods)[SPX_LANE_ELEMS],
                                                 int n_valid, int lane)
{
    // run sums: a[j] = sum of the products of the run ending at j (inside the lane)
    double a[SPX_LANE_ELEMS];
    a[0] = prods[0];
#pragma unroll
    for (int j = 1; j < SPX_LANE_ELEMS; ++j)
        a[j] = prods[j] + ((rows[j] == rows[j - 1]) ? a[j - 1] : 0.0);
    const bool active = FULL || n_valid > 0;
    const int last_row = rows[SPX_LANE_ELEMS - 1];
    const bool single = rows[0] == last_row;     // rows are never interleaved inside a lane
    const int prev_last = __shfl_up(last_row, 1);
    const bool cont = active && lane != 0 && single && prev_last == last_row;
    const uint64_t cont_mask = __ballot(cont);
    double tail = a[SPX_LANE_ELEMS - 1];
    bool tail_adds = active;
    if (__popcll(cont_mask) >= 8) {
        // segmented inclusive scan over the lanes' last runs
        int head = !cont;
        const int head0 = head;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double a2 = __shfl_up(tail, d);
            const int h2 = __shfl_up(head, d);
            if (lane >= d && !head) {
                tail += a2;
                head |= h2;
            }
        }
        const int next_head = __shfl_down(head0, 1);
        tail_adds = active && (lane == 63 || next_head);
    }
#pragma unroll
    for (int j = 0; j < SPX_LANE_ELEMS - 1; ++j) {
        const bool ends = rows[j] != rows[j + 1];
        if (ends && (FULL || j < n_valid)) atomicAdd(&tile[rows[j]], a[j]);
    }
    if (tail_adds) atomicAdd(&tile[last_row], tail);
}

// What a wavefront loads for one pass before it can decode anything: issued
// one pass ahead of the arithmetic so that the latency of the value stream
// hides behind the previous pass (software pipeline, 2 stages).
struct PassLoad {
    double2 v01, v23;     // the lane's SPX_LANE_ELEMS values
    uint4 q;              // unit pass: descriptor #(first_desc + lane)
                          // delta pass: the lane's column offsets
    uint32_t nib;         // the lane's segment-start bits
    uint32_t rank0;       // segment starts in front of the pass
    uint32_t seg;         // delta pass: row of segment #(first_seg + lane)
    int e0;               // first nonzero of the lane inside the region
    int n_valid;
};

template <bool DELTA>
__device__ __forceinline__ PassLoad issue_pass(const KernelArgs &a, const SpxRowBlock &rb,
                                               int pass, int n_unit_passes, int lane)
{
    PassLoad L;
    const int n = DELTA ? rb.n_delta_elems : rb.n_unit_elems;
    L.e0 = pass * SPX_PASS_ELEMS + lane * SPX_LANE_ELEMS;
    L.n_valid = min(max(n - L.e0, 0), SPX_LANE_ELEMS);
    const uint32_t gpass = (rb.bits_off >> 3) + (uint32_t) (DELTA ? n_unit_passes + pass : pass);
    const uint32_t w = a.bits[gpass * SPX_PASS_WORDS + ((uint32_t) lane >> 3)];
    L.nib = (w >> ((lane & 7) * 4)) & 0xFu;
    L.rank0 = a.pass_rank[gpass];
    const double *vals = a.values + rb.val_off + (DELTA ? ((rb.n_unit_elems + 3) & ~3) : 0);
    // regions are padded to whole lanes with zeros, so whole-lane loads are safe
    L.v01 = *reinterpret_cast<const double2 *>(vals + (uint32_t) L.e0);
    L.v23 = *reinterpret_cast<const double2 *>(vals + (uint32_t) L.e0 + 2);
    L.q = make_uint4(0, 0, 0, 0);
    L.seg = 0;
    // descriptors / segment rows this pass can touch start at index
    // max(rank0 - 1, 0); each lane fetches one of the next 64 (the arrays
    // carry slack, so reading past a row-block's own entries is harmless)
    const uint32_t first = L.rank0 ? L.rank0 - 1 : 0;
    if (DELTA) {
        const uint8_t *cidx = a.cidx + rb.cidx_off;
        if (rb.cidx_width == 4) {
            L.q = *reinterpret_cast<const uint4 *>(cidx + (uint32_t) L.e0 * 4u);
        } else {
            const uint2 o = *reinterpret_cast<const uint2 *>(cidx + (uint32_t) L.e0 * 2u);
            L.q = make_uint4(o.x & 0xffffu, o.x >> 16, o.y & 0xffffu, o.y >> 16);
        }
        L.seg = a.segrows[rb.seg_off + first + (uint32_t) lane];
    } else {
        L.q = *reinterpret_cast<const uint4 *>(a.descs + rb.desc_off + first + (uint32_t) lane);
    }
    return L;
}

// decoded view of a unit descriptor, ready for stepping
struct Walk {
    int rr, cc;        // row (inside the row-block) and column of the current nonzero
    int in;            // position inside the block row (dense blocks)
    int mod;           // block row length, 0 for linear units
    int sr, sc;        // per-nonzero strides (linear), (0, 1) for blocks
};

__device__ __forceinline__ Walk walk_begin(const uint4 &q)
{
    Walk w;
    w.cc = (int) q.x;
    w.rr = (int) (q.z >> 16);
    w.mod = (int) ((q.w >> 16) & 0xffu);
    w.in = 0;
    w.sr = w.mod ? 0 : (int) (int16_t) (q.w & 0xffffu);
    w.sc = w.mod ? 1 : (int) q.y;
    return w;
}

__device__ __forceinline__ void walk_step(Walk &w)
{
    const int in1 = w.in + 1;
    const bool wrap = in1 == w.mod;          // never true for linear units (mod == 0)
    w.cc += wrap ? 1 - w.mod : w.sc;
    w.rr += wrap ? 1 : w.sr;
    w.in = wrap ? 0 : in1;
}

// Decode + multiply + reduce one pass of the unit region.
template <bool SYM, bool FULL>
__device__ __forceinline__ void compute_unit_pass(const KernelArgs &a, const SpxRowBlock &rb,
                                                  double *tile, uint4 *stage,
                                                  const PassLoad &L, int lane)
{
    // exchange the 64 prefetched descriptors through LDS
    stage[lane] = L.q;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    const uint32_t first = L.rank0 ? L.rank0 - 1 : 0;
    const uint32_t idx0 = L.rank0 + rank_before(L.nib) + (L.nib & 1u) - 1u;   // lane's first unit
    const uint32_t rel0 = idx0 - first;
    const uint4 *gdescs = reinterpret_cast<const uint4 *>(a.descs + rb.desc_off);
    // more than 64 units in reach of one pass: rare, read those from memory
    const bool far = __any((rel0 + (uint32_t) __popc(L.nib >> 1)) >= 64u);

    const uint4 q0 = (!far || rel0 < 64u) ? stage[rel0 & 63u] : gdescs[idx0];
    Walk w = walk_begin(q0);
    {
        // position of the lane's first nonzero inside its unit
        const int k = L.e0 - (int) (q0.z & 0xffffu);
        if (w.mod) {
            const int out = (int) (((float) k + 0.5f) * __frcp_rn((float) w.mod));
            w.in = k - out * w.mod;
            w.rr += out;
            w.cc += w.in;
        } else {
            w.rr += k * w.sr;
            w.cc += k * w.sc;
        }
    }
    const double v[SPX_LANE_ELEMS] = {L.v01.x, L.v01.y, L.v23.x, L.v23.y};
    int rows[SPX_LANE_ELEMS];
    double prods[SPX_LANE_ELEMS];
    const bool inner_starts = __any(L.nib & 0xEu);
#pragma unroll
    for (int j = 0; j < SPX_LANE_ELEMS; ++j) {
        if (j > 0) {
            walk_step(w);
            if (inner_starts) {
                // a new unit may start at this nonzero: fetch its descriptor
                // for every lane and select (no divergent branch)
                const uint32_t rel = rel0 + (uint32_t) __popc(L.nib & ((2u << j) - 2u));
                const uint4 qj = (!far || rel < 64u) ? stage[rel & 63u] : gdescs[first + rel];
                const Walk nw = walk_begin(qj);
                const bool fresh = (L.nib >> j) & 1u;
                w.rr = fresh ? nw.rr : w.rr;
                w.cc = fresh ? nw.cc : w.cc;
                w.in = fresh ? 0 : w.in;
                w.mod = fresh ? nw.mod : w.mod;
                w.sr = fresh ? nw.sr : w.sr;
                w.sc = fresh ? nw.sc : w.sc;
            }
        }
        const bool ok = FULL || j < L.n_valid;
        // padding keeps the previous row (it merges into that run with a zero)
        rows[j] = ok ? w.rr : (j ? rows[j - 1] : 0);
        const uint32_t c = ok ? (uint32_t) w.cc : 0u;
        const double xv = (a.ablate & 2u) ? 1.0 : a.x[c];
        prods[j] = ok ? v[j] * xv : 0.0;
        if (SYM && ok) atomicAdd(&a.y[c], a.alpha * v[j] * a.x[rb.row0 + (uint32_t) w.rr]);
    }
    if (a.ablate & 1u) {
        const double sacc = prods[0] + prods[1] + prods[2] + prods[3];
        if (sacc == 123.456) tile[0] = sacc;
        return;
    }
    reduce_into_tile<FULL>(tile, rows, prods, L.n_valid, lane);
}

// Same for the delta region (leftover nonzeros, row-major).
template <bool SYM, bool FULL>
__device__ __forceinline__ void compute_delta_pass(const KernelArgs &a, const SpxRowBlock &rb,
                                                   double *tile, uint4 *stage,
                                                   const PassLoad &L, int lane)
{
    uint32_t *stage32 = reinterpret_cast<uint32_t *>(stage);
    stage32[lane] = L.seg;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    const uint32_t first = L.rank0 ? L.rank0 - 1 : 0;
    const uint32_t idx0 = L.rank0 + rank_before(L.nib) + (L.nib & 1u) - 1u;
    const uint32_t rel0 = idx0 - first;
    const uint16_t *segrows = a.segrows + rb.seg_off;
    const bool far = __any((rel0 + (uint32_t) __popc(L.nib >> 1)) >= 64u);

    const double v[SPX_LANE_ELEMS] = {L.v01.x, L.v01.y, L.v23.x, L.v23.y};
    const uint32_t off[SPX_LANE_ELEMS] = {L.q.x, L.q.y, L.q.z, L.q.w};
    int rows[SPX_LANE_ELEMS];
    double prods[SPX_LANE_ELEMS];
#pragma unroll
    for (int j = 0; j < SPX_LANE_ELEMS; ++j) {
        const uint32_t rel = rel0 + (uint32_t) __popc(L.nib & ((2u << j) - 2u));
        const int r = (!far || rel < 64u) ? (int) stage32[rel & 63u] : (int) segrows[first + rel];
        const bool ok = FULL || j < L.n_valid;
        rows[j] = ok ? r : (j ? rows[j - 1] : 0);
        const uint32_t c = ok ? rb.cbase + off[j] : 0u;
        const double xv = (a.ablate & 2u) ? 1.0 : a.x[c];
        prods[j] = ok ? v[j] * xv : 0.0;
        if (SYM && ok) atomicAdd(&a.y[c], a.alpha * v[j] * a.x[rb.row0 + (uint32_t) r]);
    }
    if (a.ablate & 1u) {
        const double sacc = prods[0] + prods[1] + prods[2] + prods[3];
        if (sacc == 123.456) tile[0] = sacc;
        return;
    }
    reduce_into_tile<FULL>(tile, rows, prods, L.n_valid, lane);
}

template <bool SYM>
__device__ __forceinline__ void compute_pass(const KernelArgs &a, const SpxRowBlock &rb,
                                             double *tile, uint4 *stage, const PassLoad &L,
                                             int t, int n_unit_passes, int lane)
{
    // all lanes hold SPX_LANE_ELEMS real nonzeros except in a region's last pass
    const bool full = __all(L.n_valid == SPX_LANE_ELEMS);
    if (t < n_unit_passes) {
        if (full) compute_unit_pass<SYM, true>(a, rb, tile, stage, L, lane);
        else compute_unit_pass<SYM, false>(a, rb, tile, stage, L, lane);
    } else {
        if (full) compute_delta_pass<SYM, true>(a, rb, tile, stage, L, lane);
        else compute_delta_pass<SYM, false>(a, rb, tile, stage, L, lane);
    }
}

// One workgroup owns one row-block; its wavefronts take the passes in turn
// (wave w: passes w, w+4, ...), loading pass t+4 while computing pass t, and
// accumulate into one y tile in LDS, which is written out at the end.
template <bool SYM>
__global__ __launch_bounds__(BLOCK_THREADS)
void csx_spmv_kernel(KernelArgs a)
{
    __shared__ double tile[SPX_MAX_RB_ROWS];
    __shared__ uint4 stage_all[WAVES_PER_BLOCK][64];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware order: workgroup b runs on XCD b % 8; give each XCD one
    // contiguous eighth of the row-blocks (gridDim.x is a multiple of 8)
    const uint32_t nb = gridDim.x;
    const uint32_t rb_idx = (blockIdx.x & 7u) * (nb >> 3) + (blockIdx.x >> 3);
    if (rb_idx >= a.n_rb) return;

    const SpxRowBlock rb = a.rbs[rb_idx];
    const int n_rows = rb.n_rows;
    const int n_unit_passes = (rb.n_unit_elems + SPX_PASS_ELEMS - 1) / SPX_PASS_ELEMS;
    const int n_delta_passes = (rb.n_delta_elems + SPX_PASS_ELEMS - 1) / SPX_PASS_ELEMS;
    const int n_pass = n_unit_passes + n_delta_passes;
    uint4 *stage = stage_all[wave];

    // first loads go out before the tile is even zeroed
    int t = wave;
    PassLoad cur;
    if (t < n_pass)
        cur = (t < n_unit_passes) ? issue_pass<false>(a, rb, t, n_unit_passes, lane)
                                  : issue_pass<true>(a, rb, t - n_unit_passes, n_unit_passes, lane);
    for (int i = threadIdx.x; i < n_rows; i += BLOCK_THREADS) tile[i] = 0.0;
    __syncthreads();

    while (t < n_pass) {
        const int tn = t + WAVES_PER_BLOCK;
        PassLoad nxt;
        if (tn < n_pass)
            nxt = (tn < n_unit_passes)
                      ? issue_pass<false>(a, rb, tn, n_unit_passes, lane)
                      : issue_pass<true>(a, rb, tn - n_unit_passes, n_unit_passes, lane);
        compute_pass<SYM>(a, rb, tile, stage, cur, t, n_unit_passes, lane);
        cur = nxt;
        t = tn;
    }
    __syncthreads();

    // ---------------- write the owned rows ------------------------------------------------
    if (SYM) {
        for (int i = threadIdx.x; i < n_rows; i += BLOCK_THREADS)
            atomicAdd(&a.y[rb.row0 + i], a.alpha * tile[i]);
    } else if (rb.flags & SPX_RB_SHARED) {
        if (threadIdx.x == 0) a.carry[rb.carry_slot] = tile[0];
    } else if (a.beta == 0.0) {
        for (int i = threadIdx.x; i < n_rows; i += BLOCK_THREADS)
            a.y[rb.row0 + i] = a.alpha * tile[i];
    } else {
        for (int i = threadIdx.x; i < n_rows; i += BLOCK_THREADS) {
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
    uint16_t *pass_rank = nullptr;
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
    m->descs = upload(s.descs, 72);
    m->bits = upload(s.bits, SPX_PASS_WORDS);
    m->pass_rank = upload(s.pass_rank, 4);
    m->cidx = upload(s.cidx, 64);
    m->segrows = upload(s.segrows, 80);
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
    (void) hipFree(m->bits); (void) hipFree(m->pass_rank);
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
    a.pass_rank = m->pass_rank;
    a.cidx = m->cidx; a.segrows = m->segrows; a.x = d_x; a.y = d_y;
    a.carry = m->carry; a.alpha = alpha; a.beta = beta; a.n_rb = m->n_rb;
    static const char *abl = getenv("SPX_ABLATE");
    a.ablate = abl ? (uint32_t) atoi(abl) : 0u;
    uint32_t blocks = (m->n_rb + 7u) & ~7u;
    if (m->symmetric) {
        const int t = 256;
        hipLaunchKernelGGL(csx_sym_init_kernel, dim3((unsigned)((m->nrows + t - 1) / t)),
                           dim3(t), 0, stream, d_y, d_x, m->dvalues, m->nrows,
                           m->own_lo, m->own_hi, alpha, beta);
        if (blocks)
            hipLaunchKernelGGL(csx_spmv_kernel<true>, dim3(blocks),
                               dim3(BLOCK_THREADS), 0, stream, a);
    } else {
        if (blocks)
            hipLaunchKernelGGL(csx_spmv_kernel<false>, dim3(blocks),
                               dim3(BLOCK_THREADS), 0, stream, a);
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
