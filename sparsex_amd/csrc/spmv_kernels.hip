// spmv_kernels.hip -- the CSX interpreter for gfx950 (MI355X).
//
// One workgroup walks one row-block of the descriptor stream (gpu_format.h).
// A pass is 64 row segments of equal width, one per lane: contiguous 16-byte
// per-lane reads of the interleaved values, the pass' segment-start mask
// ranked with mbcnt to find each lane's unit descriptor, strided decode of
// (row, first column), x gathered through L2, W fused multiply-adds per lane,
// one LDS add per lane into the row-block's y tile, and one coalesced write
// of the owned rows of y.  Leftover nonzeros (CSX delta units) run through the
// same code as gather passes: a lane owns up to 8 leftovers of one row, each
// with its own column offset.
//
// Semantics restated from the reference's SpMV templates
// (src/templates/csx_spmv_tmpl.c:66-101 and the per-unit bodies
// delta/horiz/vert/diag/rdiag/block_row/block_col _tmpl.c; symmetric:
// csx_sym_spmv_tmpl.c:60-106): every stored nonzero a(r,c) contributes
// alpha*a*x[c] to y[r]; on the symmetric path the stream also holds the mirror
// image of every stored unit, so a(r,c) contributes alpha*a*x[r] to y[c] too.
#include "device.hpp"
#include "stream_index.hpp"
#include "threads.hpp"
#include "spmv_device.hpp"
#include "spmv_sym_device.hpp"
#include "spx_abl.hpp"
#include "xwindows.hpp"
#include "sxplan.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace spx {

// spmv_sx_kernels.hip
void launch_spmv_sx(int waves, unsigned blocks, size_t lds_bytes, void *stream, const KernelArgs &a, const XcdSplit &xs,
                    const uint32_t *sx_tab);
size_t spmv_sx_header_bytes(uint32_t pass_stride);
void spmv_sx_allow_lds(size_t bytes);

// spmv_xw_kernels.hip
void launch_spmv_xw(int waves, unsigned blocks, size_t lds_bytes, void *stream, const KernelArgs &a, const XcdSplit &xs);
void spmv_xw_allow_lds(size_t bytes);

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


// One workgroup owns one row-block; its wavefronts take the passes in turn
// (wave w: passes w, w+4, ...) and accumulate into one y tile in LDS, which
// is written out (y = alpha*tile + beta*y) at the end.
// SYM: the symmetric variant with tiles (dynamic LDS: the row-block's
// transposed-sum slots in front of its y tile; the sums of columns owned by
// other row-blocks are spilled for csx_symfix_kernel).
// ATOMIC (symmetric tiles only): the row-block hands everything over with
// global_atomic_add_f64 -- its own rows (csx_sym_init_kernel has put beta*y and the
// diagonal term there) and, in aligned groups of eight, the transposed sums of
// the columns in front of it -- instead of spilling them for a second kernel.
//
// DET (spx.gpu.deterministic): every wavefront adds into a y tile (and slots) of its
// own, and the copies are summed in wavefront order before the write-out -- the
// only thing in this library whose order of additions is not fixed is the LDS adds
// of different wavefronts of a workgroup into the shared tile.
template <bool SYM, bool ATOMIC, int WAVES_PER_BLOCK, bool DET = false, bool SEGS = false, bool TILES = true>
__device__ __forceinline__ void spmv_body(const KernelArgs &a, const XcdSplit &xs,
                                          double *lds)
{
    constexpr int BLOCK_THREADS = 64 * WAVES_PER_BLOCK;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware order: workgroup b runs on XCD b % 8 and takes that XCD's next row-block
    // (the grid is 8 x the longest of the eight lists)
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t rb_idx = xs.first[xcd] + (blockIdx.x >> 3);
    if (rb_idx >= xs.first[xcd + 1u]) return;

    // the pass headers sit at a fixed stride, so the wave's first two are
    // fetched together with the row-block header, not after it
    const SpxPass *passes = a.passes + (size_t) rb_idx * a.pass_stride;
    const SpxRowBlock rb = a.rbs[rb_idx];
    SpxPass p0 = passes[wave];
    SpxPass p1 = passes[wave + WAVES_PER_BLOCK];         // (the table is padded by one stride)
    const int n_rows = rb.n_rows;
    const int n_slots = SYM ? (int) rb.n_slots : 0;
    const int core = n_slots + n_rows;                       // doubles per copy of slots + y tile
    constexpr int COPIES = DET ? WAVES_PER_BLOCK : 1;
    double *mine = lds + (DET ? wave * core : 0);            // this wavefront's slots, then its y tile
    double *tile = mine + n_slots;
    for (int i = threadIdx.x; i < COPIES * core; i += BLOCK_THREADS) lds[i] = 0.0;
    // the row-block's x window (leftovers whose columns lie close together gather
    // from LDS): staged once, coalesced
    double *win = lds + COPIES * core;
    {
        const int xw = rb.xwin_len;
        const double *xs = a.x + rb.xwin_base;
        for (int i = threadIdx.x; i < xw; i += BLOCK_THREADS) win[i] = xs[i];
    }
    // atomic hand-over: the first columns of the slot groups are fetched now, into LDS behind
    // the window, so that the hand-over at the end does not start with a load from memory
    // (a dependent L2/HBM round trip per 256 slots, at a point where the workgroup has
    // nothing else in flight)
    uint32_t *gcol_lds = reinterpret_cast<uint32_t *>(win + rb.xwin_len);
    if (ATOMIC) {
        const uint32_t *gcol = a.slot_col + (rb.spill_off >> 3);
        for (int i = threadIdx.x; i < (n_slots >> 3); i += BLOCK_THREADS) gcol_lds[i] = gcol[i];
    }
    __syncthreads();

    // wave w takes passes w, w + W, ..., two at a time when they have the same shape (they mostly
    // do: passes are sorted by width), so that their loads overlap
    const int n_pass = rb.n_pass;
    int t = wave;
    // (symmetric tiles: the tile passes at the head of the wavefront's list, one round trip each)
    if (SYM && TILES && !abl::sym_no_tile_run) symtile_run<WAVES_PER_BLOCK>(a, rb, passes, n_pass, t, p0, p1, mine, tile, lane);
    for (; t < n_pass; t += 2 * WAVES_PER_BLOCK) {
        const bool two = t + WAVES_PER_BLOCK < n_pass;
        if (SEGS && (p0.kind == SPX_PASS_SYMSEG || (two && p1.kind == SPX_PASS_SYMSEG))) {
            // read-once row segments (atomic hand-over only); whatever shares the round runs on its own
            if (two && p0.kind == SPX_PASS_SYMSEG && p1.kind == SPX_PASS_SYMSEG &&
                run_symseg2(a, rb, p0, p1, mine, tile, lane)) {
                // (both done)
            } else {
                if (p0.kind == SPX_PASS_SYMSEG) run_symseg(a, rb, p0, mine, tile, lane);
                else if (TILES && p0.kind == SPX_PASS_SYMTILE) symtile_pass(a, rb, p0, mine, tile, lane);
                else run_pass(a, rb, p0, tile, win, lane);
                if (two) {
                    if (p1.kind == SPX_PASS_SYMSEG) run_symseg(a, rb, p1, mine, tile, lane);
                    else if (TILES && p1.kind == SPX_PASS_SYMTILE) symtile_pass(a, rb, p1, mine, tile, lane);
                    else run_pass(a, rb, p1, tile, win, lane);
                }
            }
        } else if (SYM && TILES && p0.kind == SPX_PASS_SYMTILE) {
            symtile_pass(a, rb, p0, mine, tile, lane);
            if (two) {
                if (p1.kind == SPX_PASS_SYMTILE) symtile_pass(a, rb, p1, mine, tile, lane);
                else run_pass(a, rb, p1, tile, win, lane);
            }
        } else if (two) {
            if (p0.kind == p1.kind && p0.width == p1.width) {
                if (p0.kind == SPX_PASS_GATHER) run_units<2, 1>(a, rb, {p0, p1}, tile, win, lane);
                else if (p0.kind == SPX_PASS_GATHER_LDS) run_units<2, 2>(a, rb, {p0, p1}, tile, win, lane);
                else run_units<2, 0>(a, rb, {p0, p1}, tile, win, lane);
            } else {
                run_pass(a, rb, p0, tile, win, lane);
                if (SYM && TILES && p1.kind == SPX_PASS_SYMTILE) symtile_pass(a, rb, p1, mine, tile, lane);
                else run_pass(a, rb, p1, tile, win, lane);
            }
        } else {
            run_pass(a, rb, p0, tile, win, lane);
        }
        if (t + 2 * WAVES_PER_BLOCK < n_pass) {
            p0 = passes[t + 2 * WAVES_PER_BLOCK];
            p1 = passes[t + 3 * WAVES_PER_BLOCK];
        }
    }
    __syncthreads();
    if (DET) {
        // the wavefronts' copies, summed in wavefront order into the first one
        for (int i = threadIdx.x; i < core; i += BLOCK_THREADS) {
            double t = lds[i];
#pragma unroll
            for (int w = 1; w < COPIES; ++w) t += lds[w * core + i];
            lds[i] = t;
        }
        __syncthreads();
        tile = lds + n_slots;
    }

    // ---------------- write the owned rows ------------------------------------------------
    if (rb.flags & SPX_RB_SHARED) {
        if (threadIdx.x == 0) a.carry[rb.carry_slot] = tile[0];
    } else if (ATOMIC) {
        if (abl::sym_no_own) {
            // (experiment build: the own rows stay where they are)
        } else if ((rb.flags & SPX_RB_PRIVATE) && a.dvalues_priv) {
            // nobody else adds to these rows (mark_private_rowblocks): stored, with the diagonal
            // term and beta * y; the init pass leaves them out
            for (int i = threadIdx.x; i < n_rows; i += BLOCK_THREADS) {
                const size_t g = (size_t) rb.row0 + i;
                double t = a.alpha * (tile[i] + a.dvalues_priv[g] * a.x[g]);
                if (a.beta_priv != 0.0) t += a.beta_priv * a.y[g];
                a.y[g] = t;
            }
        } else {
            for (int i = threadIdx.x; i < n_rows; i += BLOCK_THREADS)
                atomicAdd(&a.y[(size_t) rb.row0 + i], a.alpha * tile[i]);
        }
        if (!abl::sym_no_handover)
            for (int i = threadIdx.x; i < n_slots; i += BLOCK_THREADS)
                atomicAdd(&a.y[(size_t) gcol_lds[i >> 3] + (i & 7)], a.alpha * lds[i]);
    } else {
        for (int i = threadIdx.x; i < n_rows; i += BLOCK_THREADS) {
            const size_t g = (size_t) rb.row0 + i;
            double t = tile[i];
            if (a.dvalues) t += a.dvalues[g] * a.x[g];
            t *= a.alpha;
            if (a.beta != 0.0) t += a.beta * a.y[g];
            a.y[g] = t;
        }
    }
    if (SYM && !ATOMIC)
        for (int i = threadIdx.x; i < n_slots; i += BLOCK_THREADS) a.spill[rb.spill_off + i] = lds[i];
}


// (Individual scalar arguments, most urgent first.  Preloading them into SGPRs
// at wave launch -- hipcc -mllvm -amdgpu-kernarg-preload-count=16 -- was
// measured: it removes the kernarg fetch in front of the first real load but
// costs more at dispatch, cant 7.5 -> 8.0 us; not used.)
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES)
void csx_spmv_kernel(SPX_KERNEL_PARAMS)
{
    SPX_KERNEL_ARGS(a);
    extern __shared__ double lds_dyn[];      // y tile, then the x window
    spmv_body<false, false, WAVES>(a, xcd_split, lds_dyn);
}

// general path, column slices in one launch (SPX_RB_ACCUM): every row-block adds its y tile to
// y with global atomics (lanes of consecutive rows: 64-byte groups), on top of csx_scale_kernel
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES)
void csx_spmv_accum_kernel(SPX_KERNEL_PARAMS)
{
    SPX_KERNEL_ARGS(a);
    extern __shared__ double lds_dyn[];
    spmv_body<false, true, WAVES>(a, xcd_split, lds_dyn);
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES)
void csx_spmv_symtile_kernel(SPX_KERNEL_PARAMS)
{
    SPX_KERNEL_ARGS(a);
    extern __shared__ double lds_dyn[];
    spmv_body<true, false, WAVES>(a, xcd_split, lds_dyn);
}

// the atomic hand-over kernel for streams that also hold read-once row segments
// (SPX_PASS_SYMSEG): a kernel of its own so that the tile-only one keeps its registers
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES)
void csx_spmv_symseg_kernel(SPX_KERNEL_PARAMS)
{
    SPX_KERNEL_ARGS(a);
    extern __shared__ double lds_dyn[];
    spmv_body<true, true, WAVES, false, true>(a, xcd_split, lds_dyn);
}

// ... and the same for streams with such segments and no tiles at all (a stencil matrix)
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES)
void csx_spmv_symseg_notile_kernel(SPX_KERNEL_PARAMS)
{
    SPX_KERNEL_ARGS(a);
    extern __shared__ double lds_dyn[];
    spmv_body<true, true, WAVES, false, true, false>(a, xcd_split, lds_dyn);
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES)
void csx_spmv_det_kernel(SPX_KERNEL_PARAMS)
{
    SPX_KERNEL_ARGS(a);
    extern __shared__ double lds_dyn[];      // a y tile per wavefront, then the x window
    spmv_body<false, false, WAVES, true>(a, xcd_split, lds_dyn);
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES)
void csx_spmv_symtile_det_kernel(SPX_KERNEL_PARAMS)
{
    SPX_KERNEL_ARGS(a);
    extern __shared__ double lds_dyn[];
    spmv_body<true, false, WAVES, true>(a, xcd_split, lds_dyn);
}

// (forcing eight wavefronts per SIMD on this kernel -- amdgpu_waves_per_eu(8, 8): 64 VGPRs and
// 32 B of scratch instead of 70 -- was measured on syn-nd24k: 25.5 -> 30.5 us; not used)
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES)
void csx_spmv_symtile_atomic_kernel(SPX_KERNEL_PARAMS)
{
    SPX_KERNEL_ARGS(a);
    extern __shared__ double lds_dyn[];
    spmv_body<true, true, WAVES>(a, xcd_split, lds_dyn);
}

// symmetric tiles, second step: every row collects the transposed sums that
// other row-blocks spilled for it (fixed order: deterministic).  A wavefront
// takes eight consecutive rows, lanes (g, r) = (lane >> 3, lane & 7): row r's
// entries g, g+8, ...  The eight columns of a tile are eight consecutive rows
// here with consecutive slots, so the eight lanes of a g read one 64-byte line.
// (Measured: one thread per row 16.4 us, this 9.1 us, a workgroup per eight
// rows 9.6 us on syn-nd24k -- the kernel is bound by its ~300 k scattered
// line requests, not by the depth of its loop.)
__global__ __launch_bounds__(256)
void csx_symfix_kernel(const uint32_t *fix_ptr, const uint32_t *fix_idx,
                       const double *spill, double *y, double alpha, uint32_t nrows)
{
    const uint32_t lane = threadIdx.x & 63u;
    // XCD-aware: workgroup b runs on XCD b % 8; each XCD takes one contiguous
    // eighth of the rows, so that the two halves of a 128-byte spill line (the
    // slots of two neighbouring tile columns) are asked for by the same L2
    const uint32_t blk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const uint32_t row = (blk * 4u + (threadIdx.x >> 6)) * 8u + (lane & 7u);
    const uint32_t g = lane >> 3;
    double s = 0.0;
    if (row < nrows) {
        const uint32_t e = fix_ptr[row + 1];
        for (uint32_t k = fix_ptr[row] + g; k < e; k += 8u) s += spill[fix_idx[k]];
    }
    s += __shfl_xor(s, 8);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (g == 0 && row < nrows && s != 0.0) y[row] += alpha * s;
}

// rows split over several row-blocks: sum their partials
__global__ void csx_fixup_kernel(const SpxSharedRow *shared, uint32_t n_shared,
                                 const double *carry, double *y, double alpha,
                                 double beta, const double *dvalues, const double *x)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_shared) return;
    const SpxSharedRow sr = shared[i];
    double s = dvalues ? dvalues[sr.row] * x[sr.row] : 0.0;
    for (uint32_t k = 0; k < sr.n_slots; ++k) s += carry[sr.first_slot + k];
    y[sr.row] = (beta == 0.0) ? alpha * s : alpha * s + beta * y[sr.row];
}

// column slices in one launch, first step: y <- beta * y on the rows [lo, hi)
__global__ void csx_scale_kernel(double *y, size_t lo, size_t hi, double beta)
{
    const size_t i = lo + (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < hi) y[i] = beta == 0.0 ? 0.0 : beta * y[i];
}

// symmetric path, first step: y <- beta*y + alpha*diag(A)*x on the owned
// rows, 0 elsewhere (the main kernel then accumulates; on several GPUs the
// per-GPU vectors are summed afterwards)
__global__ void csx_sym_init_kernel(double *y, const double *x, const double *dvalues,
                                    size_t first, size_t nrows, size_t own_lo, size_t own_hi,
                                    double alpha, double beta)
{
    const size_t i = first + (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows) return;
    double v = 0.0;
    if (i >= own_lo && i < own_hi) {
        v = alpha * dvalues[i] * x[i];
        if (beta != 0.0) v += beta * y[i];
    }
    y[i] = v;
}

// symmetric slice: the thinly spread part of the mirror image on rows of other
// processes (GpuStream::mirror_*): one thread per such row, its few nonzeros in
// fixed order.  The rows are distinct and no row-block touches them.
__global__ void csx_sym_mirror_rows_kernel(const uint32_t *rows, const uint32_t *ptr, const uint32_t *col,
                                           const double *val, const double *x, double *y, double alpha,
                                           uint32_t n)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    double s = 0.0;
    for (uint32_t k = ptr[t]; k < ptr[t + 1]; ++k) s = fma(val[k], x[col[k]], s);
    y[rows[t]] = alpha * s;          // (nothing else adds to these rows: no need to clear them first)
}

// ---- host side ------------------------------------------------------------------------------

struct DeviceMatrix {
    int device = 0;
    size_t nrows = 0, ncols = 0;
    bool symmetric = false;
    bool sym_fused = false;
    uint32_t pass_stride = 1;
    size_t own_lo = 0, own_hi = 0;
    size_t init_lo = 0;       // symmetric slice with an exchange plan: first row to clear
    bool init_limited = false;   // ... and whether one is attached (spx_hip_mat_dist_attach)
    uint32_t n_rb = 0, n_shared = 0, n_carry = 0;
    SpxRowBlock *rbs = nullptr;
    double *values = nullptr;
    SpxUnitDesc *descs = nullptr;
    SpxPass *passes = nullptr;
    uint8_t *cidx = nullptr;
    uint16_t *segrows = nullptr;
    SpxSharedRow *shared = nullptr;
    double *carry = nullptr;
    double *dvalues = nullptr;
    // symmetric tiles
    bool has_tiles = false;
    int waves = 4;            // wavefronts per workgroup of the SpMV kernel (2, 4 or 8)
    int waves_req = 4;        // ... as asked for (per-wavefront tiles may lower `waves` to fit the LDS)
    uint32_t n_spill = 0, lds_doubles = SPX_MAX_RB_ROWS;
    double *spill = nullptr;
    uint32_t *fix_ptr = nullptr, *fix_idx = nullptr;
    size_t n_fix_ptr = 0, n_fix_idx = 0;
    bool sym_atomic = false;   // transposed sums go straight into y (global atomics), no second kernel
    size_t n_private_rb = 0;
    bool use_private = false;  // SPX_RB_PRIVATE honoured (few large pieces; else one init launch over everything)
    size_t max_slot_groups = 0;   // of the row-block with the most slots
    // one launch per column phase (general path; otherwise a single one over all row-blocks):
    // the row-blocks of every XCD (balanced by values) and the length of the longest list
    std::vector<XcdSplit> xcd_split;
    std::vector<uint32_t> xcd_longest;
    // launch order (stream_band_order): device row-block i is row-block launch_order[i] of the
    // stream as the host holds it (empty: the same order); band_stride: the row distance found
    bool accum = false;           // SPX_RB_ACCUM: the column slices run in one launch and add to y
    std::vector<uint32_t> launch_order;
    size_t band_stride = 0;
    bool launched_since_edit = true;      // a product was enqueued since the last set_entry (device_poke waits once)
    bool has_symsegs = false;     // the stream holds SPX_PASS_SYMSEG passes
    bool has_symtiles = false;    // ... SPX_PASS_SYMTILE passes
    bool wave_tiles = false;      // a y tile per wavefront, summed in wavefront order before the write-out
    bool deterministic = false;   // spx.gpu.deterministic: wave tiles + fixed-order hand-overs, pinned
    uint32_t *slot_col = nullptr;
    size_t n_slot_col = 0;
    // rows of SPX_RB_PRIVATE row-blocks, merged and ascending: the init pass of the atomic
    // hand-over leaves them out (empty when there are too many pieces to be worth it)
    std::vector<std::pair<size_t, size_t>> private_rows;
    // symmetric slice: thin mirror image as a CSR over rows of other processes
    uint32_t n_mirror_rows = 0;
    size_t n_mirror_nnz = 0;
    uint32_t *mirror_rows = nullptr, *mirror_ptr = nullptr, *mirror_col = nullptr;
    double *mirror_val = nullptr;
    // staging vectors of the host-pointer path
    double *d_x = nullptr, *d_y = nullptr;
    double *p_x = nullptr, *p_y = nullptr;      // pinned
    uint64_t x_version = 0;                      // contents of d_x (0: unknown)
    hipStream_t host_stream = nullptr;
    hipStream_t copy_stream = nullptr;          // the way back of y, part by part behind the product (device_spmv_host)
    std::vector<hipEvent_t> stage_events;       // one behind every piece of a staged download
    std::vector<hipEvent_t> part_events;        // one behind every part of a product whose y travels back in parts
    int host_parts = 0;                         // parts of the last product on host vectors (0: in one piece)
    bool host_x_by_need = false;                // ... and whether x went up in the order the parts needed it
    size_t host_parts_want = 0;                 // spx.rt.host_parts (0: HOST_PARTS / HOST_PARTS_X)
    size_t value_bytes = 0, index_bytes = 0;
    size_t n_values = 0, n_descs = 0, n_passes = 0, n_cidx = 0, n_segrows = 0;
    // every array of the stream lives in ONE allocation (2 MB-aligned pieces): one mapping, one
    // run of physically contiguous fragments as far as the driver can give them
    // (spx.gpu.arena=true; default: an allocation per array -- the arena changed nothing in the
    // run-to-run spread it was built to test, profiles/r04/spread.md)
    void *arena = nullptr;
    size_t arena_bytes = 0;
    // chunked launches (device_plan_chunks / device_spmv_chunk: the exchange of a row-partitioned
    // matrix overlaps with the product): work in front of every row-block, its first row
    std::vector<uint64_t> rb_upto;
    std::vector<uint32_t> rb_row0;
    // (two cuts side by side: slot 0 belongs to an attached exchange plan, slot 1 to spx_hip_matvec_parts --
    // a caller's ad-hoc cut must not change the one an overlapped step on another stream is walking)
    struct ChunkPlan {
        std::vector<XcdSplit> split;
        std::vector<uint32_t> longest;
        std::vector<size_t> bounds;
        size_t asked = 0;
        // (slot 2, general streams with rb_xneed) the order in which the host entry point runs the parts -- the one
        // that needs the fewest pieces of x not yet on the device first -- and the pieces that go up in front of each
        std::vector<uint32_t> order;
        std::vector<std::vector<uint32_t>> step_pieces;
        // (symmetric streams) the row-blocks in front of the stretch the parts cover: launched with the last part
        XcdSplit front;
        uint32_t front_longest = 0;
    };
    ChunkPlan chunks[3];                        // 0: an attached exchange plan's, 1: spx_hip_matvec_parts', 2: the host entry point's
    // unit windows of x in LDS (xwindows.hpp; plain general streams): a second set of pass headers and
    // descriptors for csx_spmv_xw_kernel, the window table, the LDS a launch needs
    SpxPass *passes_xw = nullptr;
    SpxUnitDesc *xdescs = nullptr;
    XwEntry *xw_tab = nullptr;
    uint32_t lds_doubles_xw = 0;
    uint32_t xw_budget = 0, xw_gap = 0;   // as the stream was uploaded (kept for spx_mat_save)
    bool xw_on = false;           // the product runs through csx_spmv_xw_kernel
    uint64_t xw_elems = 0, xw_unit_elems = 0, xw_staged = 0;
    size_t xw_rowblocks = 0;
    // the read-once passes pipelined (sxplan.hpp; symmetric streams of row segments without tiles): a second set
    // of pass headers for csx_spmv_sx_kernel and the number of SX passes at the head of every row-block
    SpxPass *passes_sx = nullptr;
    uint32_t *sx_tab = nullptr;
    bool sx_on = false;           // the product runs through csx_spmv_sx_kernel
    uint64_t sx_elems = 0, sx_sym_elems = 0;
    size_t sx_rowblocks = 0;
    // which pieces of x (of xneed_piece doubles) every row-block reads (stream_rowblock_xpieces): general streams
    // that the host entry point cuts into parts, whose x then travels in the order the parts need it
    std::vector<uint64_t> rb_xneed;
    size_t xneed_piece = 0;
    hipStream_t up_stream = nullptr;            // ... on a stream of its own
};

// the host-vector entry point (device_spmv_host)
constexpr size_t STAGE_PIECE = (size_t) 16 << 20;      // bytes
constexpr size_t HOST_PARTS = 8;                       // parts of a product whose y goes back part by part ...
constexpr size_t HOST_PARTS_X = 24, HOST_PARTS_X_SYM = 16;    // ... (x comes piece by piece: finer; swept on the bench matrix, profiles/r06/host_parts_sweep.txt)
constexpr size_t HOST_X_PIECE = (size_t) 4 << 20;      // ... in pieces of this many bytes
constexpr size_t HOST_PARTS_MIN_BYTES = (size_t) 32 << 20;   // ... where y is at least this large

static size_t host_parts_min_bytes()
{
    static const size_t v = getenv("SPX_HOST_PARTS_MIN_BYTES") ? (size_t) atoll(getenv("SPX_HOST_PARTS_MIN_BYTES"))
                                                               : HOST_PARTS_MIN_BYTES;             // (tests: small matrices)
    return v;
}

// doubles per piece of x: HOST_X_PIECE, or what keeps the vector within 64 of them
static size_t host_xpiece_doubles(size_t ncols)
{
    static const size_t env = getenv("SPX_HOST_XPIECE_BYTES") ? (size_t) atoll(getenv("SPX_HOST_XPIECE_BYTES")) / sizeof(double) : 0;
    size_t piece = env ? env : HOST_X_PIECE / sizeof(double);                                      // (tests: small pieces)
    piece = std::max(piece, (ncols + 63) / 64);
    return (piece + 511) & ~(size_t) 511;
}

int device_count()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// The arrays of a stream are placed together: device_upload() notes what goes where (`put`) and
// `Placement::flush` allocates once, clears the lot and copies every array to its place.
struct Placement {
    struct Item { void **dst; const void *src; size_t copy_bytes, alloc_bytes; };
    std::vector<Item> items;
    template <typename T, typename A>
    void put(T **dst, const std::vector<T, A> &v, size_t slack_elems = 0)
    {
        size_t bytes = (v.size() + slack_elems) * sizeof(T);
        if (bytes == 0) bytes = sizeof(T);
        items.push_back(Item{reinterpret_cast<void **>(dst), v.data(), v.size() * sizeof(T), bytes});
    }
    static size_t piece(size_t bytes)
    {
        const size_t a = bytes >= ((size_t) 1 << 20) ? ((size_t) 2 << 20) : 256u;
        return (bytes + a - 1) / a * a;
    }
    void flush(DeviceMatrix *m, bool arena)
    {
        if (!arena) {
            for (const Item &it : items) {
                void *d = nullptr;
                HIP_CHECK(hipMalloc(&d, it.alloc_bytes));
                *it.dst = d;
                HIP_CHECK(hipMemset(d, 0, it.alloc_bytes));
                if (it.copy_bytes) HIP_CHECK(hipMemcpy(d, it.src, it.copy_bytes, hipMemcpyHostToDevice));
            }
            return;
        }
        // large arrays first, each on a 2 MB boundary; the small ones share the tail
        size_t total = 0;
        for (const Item &it : items) total += piece(it.alloc_bytes);
        total = (total + ((size_t) 2 << 20) - 1) & ~(((size_t) 2 << 20) - 1);
        void *base = nullptr;
        HIP_CHECK(hipMalloc(&base, total));
        m->arena = base;
        m->arena_bytes = total;
        HIP_CHECK(hipMemset(base, 0, total));
        std::vector<size_t> order(items.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return items[a].alloc_bytes > items[b].alloc_bytes; });
        size_t at = 0;
        for (size_t i : order) {
            const Item &it = items[i];
            char *d = static_cast<char *>(base) + at;
            *it.dst = d;
            if (it.copy_bytes) HIP_CHECK(hipMemcpy(d, it.src, it.copy_bytes, hipMemcpyHostToDevice));
            at += piece(it.alloc_bytes);
        }
    }
};

DeviceMatrix *device_upload(const GpuStream &s, size_t nrows, size_t ncols,
                            bool symmetric, idx_t own_lo, idx_t own_hi, int device)
{
    if (device_count() <= 0) {
        log_msg(LOG_ERR, "no usable HIP device: the SpMV path of this library runs "
                "on an MI355X only (set spx.rt.host_only=true to tune without one)\n");
        throw FatalError("no HIP device");
    }
    if (device >= 0) HIP_CHECK(hipSetDevice(device));
    if (!s.pass_stride && !s.rbs.empty()) throw FatalError("descriptor stream was not finalized");
    DeviceMatrix *m = new DeviceMatrix;
    HIP_CHECK(hipGetDevice(&m->device));
    m->nrows = nrows;
    m->ncols = ncols;
    m->symmetric = symmetric;
    m->sym_fused = symmetric && s.sym_fused;
    m->pass_stride = s.pass_stride;
    m->waves = m->waves_req = (s.waves == 2 || s.waves == 8) ? (int) s.waves : 4;
    m->own_lo = (size_t) own_lo;
    m->own_hi = (size_t) own_hi;
    m->n_rb = (uint32_t) s.rbs.size();
    m->n_shared = (uint32_t) s.shared.size();
    m->n_carry = s.n_carry;
    Placement place;
    const std::vector<double> no_doubles;                 // (cleared arrays: nothing to copy)
    std::vector<double> dv;                               // host copies that must live until the flush
    std::vector<SpxRowBlock> rbs_ordered;
    std::vector<SpxPass> passes_ordered;
    place.put(&m->cidx, s.cidx, 64);
    place.put(&m->segrows, s.segrows, 80);
    place.put(&m->shared, s.shared);
    place.put(&m->carry, no_doubles, s.n_carry ? s.n_carry : 1);
    if (symmetric) {
        dv = s.dvalues;
        dv.resize(nrows, 0.0);
        place.put(&m->dvalues, dv);
    }
    m->n_spill = s.n_spill;
    m->lds_doubles = s.lds_doubles;
    for (const SpxRowBlock &rb : s.rbs)
        for (uint32_t k = 0; k < rb.n_pass && !m->has_tiles; ++k)
            m->has_tiles = s.passes[rb.pass_off + k].kind == SPX_PASS_SYMTILE ||
                           s.passes[rb.pass_off + k].kind == SPX_PASS_SYMSEG;
    for (const SpxRowBlock &rb : s.rbs)
        for (uint32_t k = 0; k < rb.n_pass && !m->has_symsegs; ++k)
            m->has_symsegs = s.passes[rb.pass_off + k].kind == SPX_PASS_SYMSEG;
    for (const SpxRowBlock &rb : s.rbs)
        for (uint32_t k = 0; k < rb.n_pass && !m->has_symtiles; ++k)
            m->has_symtiles = s.passes[rb.pass_off + k].kind == SPX_PASS_SYMTILE;
    // (n_slots + n_rows <= 3584 doubles = 28 KB: within the default dynamic LDS limit)
    if (s.n_spill) {
        place.put(&m->spill, no_doubles, s.n_spill);
        place.put(&m->fix_ptr, s.fix_ptr);
        place.put(&m->fix_idx, s.fix_idx);
        m->n_fix_ptr = s.fix_ptr.size();
        m->n_fix_idx = s.fix_idx.size();
        place.put(&m->slot_col, s.slot_group_col);
        m->n_slot_col = s.slot_group_col.size();
    }
    if (!s.mirror_rows.empty()) {
        m->n_mirror_rows = (uint32_t) s.mirror_rows.size();
        m->n_mirror_nnz = s.mirror_col.size();
        place.put(&m->mirror_rows, s.mirror_rows);
        place.put(&m->mirror_ptr, s.mirror_ptr);
        place.put(&m->mirror_col, s.mirror_col);
        place.put(&m->mirror_val, s.mirror_val);
    }
    m->sym_atomic = (s.sym_atomic || m->has_symsegs) && m->has_tiles;
    {
        std::vector<std::pair<size_t, size_t>> pr;
        for (const SpxRowBlock &rb : s.rbs)
            if (rb.flags & SPX_RB_PRIVATE) pr.emplace_back((size_t) rb.row0, (size_t) rb.row0 + rb.n_rows);
        std::sort(pr.begin(), pr.end());
        for (const auto &r : pr) {
            if (!m->private_rows.empty() && m->private_rows.back().second == r.first) m->private_rows.back().second = r.second;
            else m->private_rows.push_back(r);
        }
        m->n_private_rb = pr.size();
        // worth it where it takes a good part of the init pass away in a few pieces (every gap
        // is a launch of its own: syn-nd24k, a 25 us product, lost 4 us to nine of them)
        size_t covered = 0;
        for (const auto &r : m->private_rows) covered += r.second - r.first;
        if (m->private_rows.size() > 4 || covered * 4 < nrows) m->private_rows.clear();
        m->use_private = !m->private_rows.empty();
        for (const SpxRowBlock &rb : s.rbs) m->max_slot_groups = std::max<size_t>(m->max_slot_groups, (rb.n_slots + 7u) / 8u);
    }
    {
        // an eighth of the work of a launch to every XCD: values held (+ a constant per
        // row-block for its headers and its write-out)
        const size_t n = s.rbs.size();
        std::vector<uint64_t> upto(n + 1, 0);
        for (size_t i = 0; i < n; ++i) {
            const uint64_t end = i + 1 < n ? s.rbs[i + 1].val_off : (uint64_t) s.values.size();
            upto[i + 1] = upto[i] + (end > s.rbs[i].val_off ? end - s.rbs[i].val_off : 0) + 64u + 2u * s.rbs[i].n_rows;
        }
        m->rb_upto = upto;
        m->rb_row0.resize(n);
        for (size_t i = 0; i < n; ++i) m->rb_row0[i] = s.rbs[i].row0;
        std::vector<size_t> starts(1, 0);
        for (size_t i = 1; i < n; ++i)
            if (s.rbs[i].flags & SPX_RB_PHASE_START) starts.push_back(i);
        starts.push_back(n);
        m->accum = n > 0 && (s.rbs[0].flags & SPX_RB_ACCUM) != 0;
        const size_t K = starts.size() - 1;
        if (m->accum && (K == 2 || K == 4 || K == 8)) {
            // one launch: slice k on the XCDs [k * 8 / K, (k + 1) * 8 / K), its row-blocks dealt to
            // them in contiguous parts of equal values
            XcdSplit xs;
            const uint32_t per = (uint32_t)(8 / K);
            for (size_t k = 0; k < K; ++k) {
                const size_t lo = starts[k], hi = starts[k + 1];
                for (uint32_t j = 0; j < per; ++j) {
                    const uint64_t want = upto[lo] + (upto[hi] - upto[lo]) * j / per;
                    size_t i = (size_t)(std::lower_bound(upto.begin() + lo, upto.begin() + hi + 1, want) - upto.begin());
                    xs.first[k * per + j] = (uint32_t) std::min(std::max(i, lo), hi);
                }
            }
            xs.first[8] = (uint32_t) n;
            uint32_t longest = 0;
            for (uint32_t x = 0; x < 8; ++x) longest = std::max(longest, xs.first[x + 1] - xs.first[x]);
            m->xcd_split.push_back(xs);
            m->xcd_longest.push_back(longest);
            starts.assign(1, n);           // (nothing left for the sequential form below)
        } else if (m->accum) {
            throw FatalError("column slices for one launch: 2, 4 or 8 of them");
        }
        for (size_t ph = 0; ph + 1 < starts.size(); ++ph) {
            const size_t lo = starts[ph], hi = starts[ph + 1];
            XcdSplit xs;
            xs.first[0] = (uint32_t) lo;
            for (uint32_t x = 1; x < 8; ++x) {
                const uint64_t want = upto[lo] + (upto[hi] - upto[lo]) * x / 8;
                size_t i = (size_t)(std::lower_bound(upto.begin() + lo, upto.begin() + hi + 1, want) - upto.begin());
                i = std::min(std::max<size_t>(i, xs.first[x - 1]), hi);
                xs.first[x] = (uint32_t) i;
            }
            xs.first[8] = (uint32_t) hi;
            uint32_t longest = 0;
            for (uint32_t x = 0; x < 8; ++x) longest = std::max(longest, xs.first[x + 1] - xs.first[x]);
            m->xcd_split.push_back(xs);
            m->xcd_longest.push_back(longest);
        }
    }
    {
        // row-blocks and their pass headers go up in launch order: inside every XCD's part,
        // strips of a plane across the planes where the rows read x in recurring bands
        // (stream_band_order; spx.gpu.band_order=false leaves the stream's order)
        std::vector<uint32_t> order;
        if (s.band_order) {
            order.resize(s.rbs.size());
            for (size_t i = 0; i < order.size(); ++i) order[i] = (uint32_t) i;
            bool any = false;
            for (const XcdSplit &xs : m->xcd_split) {
                for (uint32_t x = 0; x < 8; ++x) {
                    size_t S = 0;
                    const std::vector<uint32_t> part = stream_band_order(s, xs.first[x], xs.first[x + 1], S);
                    if (part.empty()) continue;
                    std::copy(part.begin(), part.end(), order.begin() + xs.first[x]);
                    m->band_stride = S;
                    any = true;
                }
            }
            if (!any) order.clear();
        }
        if (order.empty()) {
            place.put(&m->values, s.values, 160);
            place.put(&m->descs, s.descs, 8);
            place.put(&m->rbs, s.rbs);
            place.put(&m->passes, s.passes, (size_t) s.pass_stride + 6 * MAX_WAVES_PER_BLOCK);
        } else {
            const size_t stride = s.pass_stride;
            std::vector<SpxRowBlock> &rbs = rbs_ordered;
            std::vector<SpxPass> &passes = passes_ordered;
            rbs.resize(s.rbs.size());
            passes.resize(s.passes.size());
            for (size_t i = 0; i < order.size(); ++i) {
                rbs[i] = s.rbs[order[i]];
                rbs[i].pass_off = (uint32_t)(i * stride);
                std::copy(s.passes.begin() + (size_t) order[i] * stride, s.passes.begin() + ((size_t) order[i] + 1) * stride,
                          passes.begin() + i * stride);
            }
            place.put(&m->values, s.values, 160);
            place.put(&m->descs, s.descs, 8);
            place.put(&m->rbs, rbs);
            place.put(&m->passes, passes, stride + 6 * MAX_WAVES_PER_BLOCK);
            m->launch_order.swap(order);
        }
    }
    if (m->has_symsegs && (size_t) m->lds_doubles * sizeof(double) + 8192u > 64u * 1024u) {
        // wide row-blocks with an x window on top: beyond the default dynamic LDS limit
        const int bytes = 160 * 1024;
        (void) hipFuncSetAttribute(reinterpret_cast<const void *>(&csx_spmv_symseg_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void) hipFuncSetAttribute(reinterpret_cast<const void *>(&csx_spmv_symseg_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void) hipFuncSetAttribute(reinterpret_cast<const void *>(&csx_spmv_symseg_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void) hipFuncSetAttribute(reinterpret_cast<const void *>(&csx_spmv_symseg_notile_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void) hipFuncSetAttribute(reinterpret_cast<const void *>(&csx_spmv_symseg_notile_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void) hipFuncSetAttribute(reinterpret_cast<const void *>(&csx_spmv_symseg_notile_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    }
    if (s.deterministic) device_set_deterministic(m, true);
    else if (s.wave_tiles) device_set_wave_tiles(m, true);
    m->n_values = s.values.size(); m->n_descs = s.descs.size(); m->n_passes = s.passes.size();
    m->n_cidx = s.cidx.size(); m->n_segrows = s.segrows.size();
    m->value_bytes = s.values.size() * sizeof(double);
    m->index_bytes = s.index_bytes();
    try {
        place.flush(m, s.arena);
    } catch (...) {
        device_free(m);
        throw;
    }
    // unit windows (plain general streams in stream order, one launch): planned from the stream as it
    // is, uploaded next to it; whether the product uses them is the caller's (the launch tuner's) choice
    m->xw_budget = s.xw_budget;
    m->xw_gap = s.xw_gap;
    if (s.xw_budget && !symmetric && !m->accum && m->xcd_split.size() == 1 && !s.rbs.empty()) {
        try {
            XwPlan plan;
            plan_unit_xwindows(s, ncols, s.xw_budget, s.xw_gap, plan, host_threads());
            if (!m->launch_order.empty()) {
                // (row-blocks and their headers went up in launch order: the window table and the headers follow;
                // the descriptors stay where they are -- a row-block finds them through its desc_off)
                const size_t stride = s.pass_stride;
                std::vector<SpxPass> po(plan.passes.size());
                std::vector<XwEntry> to(plan.tab.size());
                for (size_t i = 0; i < m->launch_order.size(); ++i) {
                    const size_t from = m->launch_order[i];
                    std::copy(plan.passes.begin() + from * stride, plan.passes.begin() + (from + 1) * stride, po.begin() + i * stride);
                    std::copy(plan.tab.begin() + from * XW_TAB, plan.tab.begin() + (from + 1) * XW_TAB, to.begin() + i * XW_TAB);
                }
                plan.passes.swap(po);
                plan.tab.swap(to);
            }
            if (plan.n_rb_windows && (size_t) plan.lds_doubles * sizeof(double) > 160u * 1024u) {
                // (a budget and row-blocks so large that a workgroup would not fit a CU's LDS: the plain kernel runs)
                log_msg(LOG_INFO, "unit windows: %u KB of LDS per workgroup do not fit, not used\n",
                        (unsigned) ((size_t) plan.lds_doubles * sizeof(double) / 1024u));
            } else if (plan.n_rb_windows) {
                auto up = [&](auto **dst, const auto &v, size_t slack) {
                    typedef typename std::remove_reference<decltype(v[0])>::type T;
                    const size_t bytes = (v.size() + slack) * sizeof(T);
                    HIP_CHECK(hipMalloc(reinterpret_cast<void **>(dst), bytes));
                    HIP_CHECK(hipMemset(*dst, 0, bytes));
                    HIP_CHECK(hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
                };
                up(&m->passes_xw, plan.passes, (size_t) s.pass_stride + 6 * MAX_WAVES_PER_BLOCK);
                up(&m->xdescs, plan.xdescs, 8);
                up(&m->xw_tab, plan.tab, 0);
                m->lds_doubles_xw = plan.lds_doubles;
                m->xw_elems = plan.unit_elems_lds;
                m->xw_unit_elems = plan.unit_elems;
                m->xw_staged = plan.staged_doubles;
                m->xw_rowblocks = plan.n_rb_windows;
                m->xw_on = s.xw_on;
                if ((size_t) m->lds_doubles_xw * sizeof(double) > 64u * 1024u) spmv_xw_allow_lds(160u * 1024u);
                log_msg(LOG_INFO, "unit windows: %zu of %zu row-blocks, %.1f %% of the unit nonzeros read x from LDS, "
                        "%.2f doubles staged per such nonzero, %u KB of LDS per workgroup\n", plan.n_rb_windows, plan.n_rb_units,
                        100.0 * (double) plan.unit_elems_lds / (double) std::max<uint64_t>(plan.unit_elems, 1),
                        (double) plan.staged_doubles / (double) std::max<uint64_t>(plan.unit_elems_lds, 1),
                        (unsigned) (m->lds_doubles_xw * sizeof(double) / 1024u));
            }
        } catch (...) {
            device_free(m);
            throw;
        }
    }
    // the read-once passes pipelined (symmetric streams of row segments, no tiles, stream order, one launch)
    if (symmetric && s.sx_plan && m->has_symsegs && !m->has_symtiles &&
        m->xcd_split.size() == 1 && !s.rbs.empty() && !s.deterministic && !s.wave_tiles) {
        try {
            SxPlan plan;
            plan_sym_pipeline(s, plan, host_threads());
            const size_t lds_need = (size_t) m->lds_doubles * sizeof(double) + m->max_slot_groups * sizeof(uint32_t) +
                                    spmv_sx_header_bytes(s.pass_stride);
            if (!m->launch_order.empty()) {
                // (row-blocks and their headers went up in launch order: the plan follows)
                const size_t stride = s.pass_stride;
                std::vector<SpxPass> po(plan.passes.size());
                std::vector<uint32_t> no(plan.n_sx.size());
                for (size_t i = 0; i < m->launch_order.size(); ++i) {
                    std::copy(plan.passes.begin() + (size_t) m->launch_order[i] * stride,
                              plan.passes.begin() + ((size_t) m->launch_order[i] + 1) * stride, po.begin() + i * stride);
                    no[i] = plan.n_sx[m->launch_order[i]];
                }
                plan.passes.swap(po);
                plan.n_sx.swap(no);
            }
            if (plan.n_rb_sx && lds_need <= 160u * 1024u) {
                const size_t pbytes = (plan.passes.size() + (size_t) s.pass_stride + 6 * MAX_WAVES_PER_BLOCK) * sizeof(SpxPass);
                HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&m->passes_sx), pbytes));
                HIP_CHECK(hipMemset(m->passes_sx, 0, pbytes));
                HIP_CHECK(hipMemcpy(m->passes_sx, plan.passes.data(), plan.passes.size() * sizeof(SpxPass), hipMemcpyHostToDevice));
                HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&m->sx_tab), plan.n_sx.size() * sizeof(uint32_t)));
                HIP_CHECK(hipMemcpy(m->sx_tab, plan.n_sx.data(), plan.n_sx.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
                m->sx_elems = plan.sx_elems;
                m->sx_sym_elems = plan.sym_elems;
                m->sx_rowblocks = plan.n_rb_sx;
                m->sx_on = s.sx_on;
                if (lds_need > 64u * 1024u) spmv_sx_allow_lds(160u * 1024u);
                log_msg(LOG_INFO, "read-once pipeline: %zu row-blocks, %llu of %llu read-once passes (%.1f %% of their nonzeros) "
                        "carry their geometry in the header\n", plan.n_rb_sx, (unsigned long long) plan.sx_passes,
                        (unsigned long long) plan.sym_passes,
                        100.0 * (double) plan.sx_elems / (double) std::max<uint64_t>(plan.sym_elems, 1));
            }
        } catch (...) {
            device_free(m);
            throw;
        }
    }
    // which pieces of x every row-block reads, where the host entry point can cut the stream into parts
    // (device_plan_chunks' conditions): device_spmv_host sends x in the order the parts need it
    if (!m->accum && !m->n_shared && m->xcd_split.size() == 1 && m->launch_order.empty() &&
        (!symmetric || !m->n_mirror_rows) && m->n_rb >= 64 &&
        m->nrows * sizeof(double) >= host_parts_min_bytes()) {
        try {
            m->xneed_piece = host_xpiece_doubles(m->ncols);
            stream_rowblock_xpieces(s, m->ncols, m->xneed_piece, m->rb_xneed, host_threads());
        } catch (...) {
            device_free(m);
            throw;
        }
    }
    return m;
}

void device_free(DeviceMatrix *m)
{
    if (!m) return;
    if (m->arena) {
        (void) hipFree(m->arena);
    } else {
        (void) hipFree(m->rbs); (void) hipFree(m->values); (void) hipFree(m->descs);
        (void) hipFree(m->passes);
        (void) hipFree(m->cidx); (void) hipFree(m->segrows); (void) hipFree(m->shared);
        (void) hipFree(m->carry);
        if (m->dvalues) (void) hipFree(m->dvalues);
        if (m->spill) (void) hipFree(m->spill);
        if (m->fix_ptr) (void) hipFree(m->fix_ptr);
        if (m->fix_idx) (void) hipFree(m->fix_idx);
        if (m->slot_col) (void) hipFree(m->slot_col);
        if (m->mirror_rows) (void) hipFree(m->mirror_rows);
        if (m->mirror_ptr) (void) hipFree(m->mirror_ptr);
        if (m->mirror_col) (void) hipFree(m->mirror_col);
        if (m->mirror_val) (void) hipFree(m->mirror_val);
    }
    if (m->passes_sx) (void) hipFree(m->passes_sx);
    if (m->sx_tab) (void) hipFree(m->sx_tab);
    if (m->passes_xw) (void) hipFree(m->passes_xw);
    if (m->xdescs) (void) hipFree(m->xdescs);
    if (m->xw_tab) (void) hipFree(m->xw_tab);
    if (m->d_x) (void) hipFree(m->d_x);
    if (m->d_y) (void) hipFree(m->d_y);
    if (m->p_x) (void) hipHostFree(m->p_x);
    if (m->p_y) (void) hipHostFree(m->p_y);
    if (m->host_stream) (void) hipStreamDestroy(m->host_stream);
    if (m->copy_stream) (void) hipStreamDestroy(m->copy_stream);
    if (m->up_stream) (void) hipStreamDestroy(m->up_stream);
    for (hipEvent_t e : m->stage_events) (void) hipEventDestroy(e);
    for (hipEvent_t e : m->part_events) (void) hipEventDestroy(e);
    delete m;
}

// `part`: the product over ONE part of the row-blocks (symmetric streams cut by device_plan_chunks): the init pass
// runs in front of the first part only, what follows the row-blocks behind the last one only
struct SpmvPart {
    XcdSplit split;
    uint32_t longest;
    bool first, last;
};
static void device_spmv_impl(DeviceMatrix *m, double alpha, const double *d_x, double beta, double *d_y, void *stream_,
                             const SpmvPart *part);

void device_spmv(DeviceMatrix *m, double alpha, const double *d_x, double beta,
                 double *d_y, void *stream_)
{
    device_spmv_impl(m, alpha, d_x, beta, d_y, stream_, nullptr);
}

static void device_spmv_impl(DeviceMatrix *m, double alpha, const double *d_x, double beta, double *d_y, void *stream_,
                             const SpmvPart *part)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != m->device)
        throw FatalError("the matrix lives on HIP device " + std::to_string(m->device) +
                         ", the calling thread's current device is " + std::to_string(cur));
    m->launched_since_edit = true;
    KernelArgs a;
    a.rbs = m->rbs; a.values = m->values; a.descs = m->descs; a.passes = m->passes;
    a.cidx = m->cidx; a.segrows = m->segrows; a.x = d_x; a.y = d_y;
    a.carry = m->carry; a.alpha = alpha; a.beta = beta; a.n_rb = m->n_rb;
    // (atomic hand-over of the tiles' sums: every row may be added to by several
    // workgroups, so beta*y and the diagonal term are put there first, as for a
    // process that holds a slice)
    const bool fused = m->sym_fused && !m->sym_atomic;
    a.dvalues = fused ? m->dvalues : nullptr;
    a.pass_stride = m->pass_stride;
    a.slot_col = m->slot_col;
    a.dvalues_priv = nullptr;      // (set where the atomic hand-over honours SPX_RB_PRIVATE)
    a.beta_priv = 0.0;

    uint32_t blocks = 0;
    XcdSplit xcd_now;
    const size_t n_launch = m->xcd_split.size();
    if (m->symmetric && !fused) {
        // y <- beta*y + alpha*diag*x on the owned rows, 0 elsewhere; the
        // row-blocks (stored lower triangle and its mirror image) then
        // accumulate on top of that
        // (attached to an exchange plan: only [init_lo, own_hi) -- the rows this
        // process owns or adds to -- are anybody's business)
        const int t = 256;
        const size_t first = m->init_limited ? m->init_lo : 0, last = m->init_limited ? m->own_hi : m->nrows;
        auto init_rows = [&](size_t lo, size_t hi) {
            if (hi > lo && !abl::sym_no_init && (!part || part->first))
                hipLaunchKernelGGL(csx_sym_init_kernel, dim3((unsigned)((hi - lo + t - 1) / t)),
                                   dim3(t), 0, stream, d_y, d_x, m->dvalues, lo, hi,
                                   m->own_lo, m->own_hi, alpha, beta);
        };
        if (m->sym_atomic && !m->wave_tiles && m->use_private && !abl::sym_no_private) {
            // (row-blocks that nobody else adds to store their rows themselves: SPX_RB_PRIVATE)
            size_t at = first;
            for (const auto &r : m->private_rows) {
                if (r.second <= at) continue;
                if (r.first >= last) break;
                init_rows(at, std::min(std::max(r.first, at), last));
                at = std::max(at, r.second);
            }
            init_rows(at, last);
            a.dvalues_priv = m->dvalues;
            a.beta_priv = beta;
        }
        else
            init_rows(first, last);
        // the thin mirror list stores its rows; whatever else lands on them (spilled tile
        // sums) is added afterwards
        if (m->n_mirror_rows && (!part || part->first))
            hipLaunchKernelGGL(csx_sym_mirror_rows_kernel, dim3((m->n_mirror_rows + 255) / 256), dim3(256), 0,
                               stream, m->mirror_rows, m->mirror_ptr, m->mirror_col, m->mirror_val, d_x, d_y,
                               alpha, m->n_mirror_rows);
        a.beta = beta = 1.0;
    }
    if (m->accum && !m->symmetric) {
        // column slices in one launch: beta * y first, every row-block adds on top
        const int t = 256;
        const size_t lo = m->own_lo, hi = m->own_hi;
        if (hi > lo)
            hipLaunchKernelGGL(csx_scale_kernel, dim3((unsigned)((hi - lo + t - 1) / t)), dim3(t), 0, stream, d_y, lo, hi, beta);
        a.beta = beta = 1.0;
    }
    a.spill = m->spill;
#define SPX_LAUNCH(KERNEL, W, LDS)                                                               \
    hipLaunchKernelGGL(KERNEL<W>, dim3(blocks), dim3(64 * W), LDS, stream, a.rbs, a.passes,      \
                       a.n_rb, a.pass_stride, xcd_now, a.values, a.descs, a.cidx,                     \
                       a.segrows, a.x, a.y, a.carry, a.dvalues, a.spill, a.slot_col, a.alpha, a.beta, \
                       a.dvalues_priv, a.beta_priv)
    bool need_symfix = false;
    for (size_t ph = 0; ph < n_launch; ++ph) {
        // (column phases: slice k > 0 adds to what the slices in front of it stored)
        xcd_now = part ? part->split : m->xcd_split[ph];
        blocks = 8u * (part ? part->longest : m->xcd_longest[ph]);
        if (ph > 0) a.beta = 1.0;
        if (blocks && m->wave_tiles && !m->accum) {
            // a copy of slots + y tile per wavefront: as many wavefronts as fit the LDS
            const int w = m->waves;
            const size_t lds = (size_t) w * m->lds_doubles * sizeof(double);
            if (m->has_tiles) {
                if (w == 2) SPX_LAUNCH(csx_spmv_symtile_det_kernel, 2, lds);
                else if (w == 8) SPX_LAUNCH(csx_spmv_symtile_det_kernel, 8, lds);
                else SPX_LAUNCH(csx_spmv_symtile_det_kernel, 4, lds);
                need_symfix = m->n_spill != 0;
            } else {
                if (w == 2) SPX_LAUNCH(csx_spmv_det_kernel, 2, lds);
                else if (w == 8) SPX_LAUNCH(csx_spmv_det_kernel, 8, lds);
                else SPX_LAUNCH(csx_spmv_det_kernel, 4, lds);
            }
        } else if (blocks && m->has_tiles) {
            // symmetric tiles: slots + y tile in dynamic LDS, then the rows collect
            // what other row-blocks spilled for them
            const size_t lds = m->lds_doubles * sizeof(double);
            const size_t lds_a = lds + m->max_slot_groups * sizeof(uint32_t);   // + the slot groups' columns
            if (m->sym_atomic && m->has_symsegs && !m->has_symtiles && m->sx_on && m->passes_sx) {
                KernelArgs as = a;
                as.passes = m->passes_sx;
                launch_spmv_sx(m->waves, blocks, lds_a + spmv_sx_header_bytes(m->pass_stride), stream, as, xcd_now, m->sx_tab);
            } else if (m->sym_atomic && m->has_symsegs && !m->has_symtiles) {
                // (16 wavefronts per workgroup, so that 2048-row row-blocks keep the SIMDs full, were
                // measured: 0.90 ms against 0.835 with 8, syn-nlpkkt; not built)
                if (m->waves == 2) SPX_LAUNCH(csx_spmv_symseg_notile_kernel, 2, lds_a);
                else if (m->waves == 8) SPX_LAUNCH(csx_spmv_symseg_notile_kernel, 8, lds_a);
                else SPX_LAUNCH(csx_spmv_symseg_notile_kernel, 4, lds_a);
            } else if (m->sym_atomic && m->has_symsegs) {
                if (m->waves == 2) SPX_LAUNCH(csx_spmv_symseg_kernel, 2, lds_a);
                else if (m->waves == 8) SPX_LAUNCH(csx_spmv_symseg_kernel, 8, lds_a);
                else SPX_LAUNCH(csx_spmv_symseg_kernel, 4, lds_a);
            } else if (m->sym_atomic) {
                if (m->waves == 2) SPX_LAUNCH(csx_spmv_symtile_atomic_kernel, 2, lds_a);
                else if (m->waves == 8) SPX_LAUNCH(csx_spmv_symtile_atomic_kernel, 8, lds_a);
                else SPX_LAUNCH(csx_spmv_symtile_atomic_kernel, 4, lds_a);
            } else if (m->waves == 2) SPX_LAUNCH(csx_spmv_symtile_kernel, 2, lds);
            else if (m->waves == 8) SPX_LAUNCH(csx_spmv_symtile_kernel, 8, lds);
            else SPX_LAUNCH(csx_spmv_symtile_kernel, 4, lds);
            need_symfix = m->n_spill && !m->sym_atomic;
        } else if (blocks && m->accum) {
            const size_t lds = m->lds_doubles * sizeof(double);
            if (m->waves == 2) SPX_LAUNCH(csx_spmv_accum_kernel, 2, lds);
            else if (m->waves == 8) SPX_LAUNCH(csx_spmv_accum_kernel, 8, lds);
            else SPX_LAUNCH(csx_spmv_accum_kernel, 4, lds);
        } else if (blocks && m->xw_on && m->passes_xw) {
            KernelArgs ax = a;
            ax.passes = m->passes_xw;
            ax.descs = m->xdescs;
            ax.xw_tab = m->xw_tab;
            launch_spmv_xw(m->waves, blocks, (size_t) m->lds_doubles_xw * sizeof(double), stream, ax, xcd_now);
        } else if (blocks) {
            const size_t lds = m->lds_doubles * sizeof(double);
            if (m->waves == 2) SPX_LAUNCH(csx_spmv_kernel, 2, lds);
            else if (m->waves == 8) SPX_LAUNCH(csx_spmv_kernel, 8, lds);
            else SPX_LAUNCH(csx_spmv_kernel, 4, lds);
        }
    }
#undef SPX_LAUNCH
    if (part && !part->last) {
        HIP_CHECK(hipGetLastError());
        return;
    }
    if (m->n_shared)
        hipLaunchKernelGGL(csx_fixup_kernel, dim3((m->n_shared + 63) / 64), dim3(64), 0,
                           stream, m->shared, m->n_shared, m->carry, d_y, alpha, beta,
                           a.dvalues, d_x);
    // the spilled column sums are added last: a row that is split over several row-blocks gets its
    // value (beta*y, the diagonal term, its partial sums) from the fix-up kernel above, by a store
    if (need_symfix)
        hipLaunchKernelGGL(csx_symfix_kernel, dim3((unsigned)((((m->nrows + 31) / 32) + 7) & ~(size_t) 7)), dim3(256),
                           0, stream, m->fix_ptr, m->fix_idx, m->spill, d_y, alpha,
                           (uint32_t) m->nrows);
    HIP_CHECK(hipGetLastError());
}

// ---- the product in K launches over consecutive parts of the row-blocks ---------------------------
// (general path, plain stream: one launch phase, no column slices, no rows split over row-blocks,
// stream order = row order).  Returns the number of parts (0: this stream cannot be cut) and the
// first row of every part (+ the end) in `row_bounds`.
size_t device_plan_chunks(DeviceMatrix *m, size_t K, std::vector<size_t> &row_bounds, int slot)
{
    DeviceMatrix::ChunkPlan &cp = m->chunks[slot < 0 || slot > 2 ? 0 : slot];
    if (K >= 2 && K == cp.asked && !cp.split.empty()) {      // (the same cut as last time)
        row_bounds = cp.bounds;
        return cp.split.size();
    }
    row_bounds.clear();
    cp.split.clear();
    cp.longest.clear();
    cp.order.clear();
    cp.step_pieces.clear();
    cp.asked = K;
    const size_t n = m->n_rb;
    if (m->accum || m->n_shared || m->xcd_split.size() != 1 || !m->launch_order.empty() ||
        n < 64 || K < 2 || m->rb_upto.size() != n + 1)
        return 0;
    // A symmetric stream can be cut where EVERY row-block stores its own rows and nobody else adds to them (one
    // stretch of SPX_RB_PRIVATE rows, the atomic hand-over: a KKT system's multiplier rows): such rows are final when
    // their row-block has run, whatever the later ones hand over to rows elsewhere.  The parts then cover that
    // stretch only; the rows outside it -- which receive sums until the last row-block has run -- follow at the end.
    if (m->symmetric &&
        !(slot == 2 && m->sym_atomic && !m->wave_tiles && m->use_private && m->private_rows.size() == 1 &&
          !m->n_mirror_rows && !m->init_limited && m->own_lo == 0 && m->own_hi == m->nrows))
        return 0;
    for (size_t i = 1; i < n; ++i)
        if (m->rb_row0[i] < m->rb_row0[i - 1]) return 0;
    size_t base = 0;
    cp.front_longest = 0;
    if (m->symmetric) {
        // (the stretch of rows that store themselves must be the tail of the row-blocks: the parts cover it; whatever
        // row-blocks lie in front of it -- rows that others add to, which travel at the end anyway -- run with the
        // LAST part: on a KKT system they are the state rows that couple with every multiplier, and in the first
        // part they would hold up its product until all of x has arrived)
        const size_t p_lo = m->private_rows[0].first;
        const size_t i0 = (size_t) (std::lower_bound(m->rb_row0.begin(), m->rb_row0.end(), (uint32_t) p_lo) - m->rb_row0.begin());
        if (i0 >= n || m->rb_row0[i0] != p_lo || n - i0 != m->n_private_rb) return 0;
        base = i0;
    }
    K = std::min<size_t>(K, (n - base) / 32);
    if (K < 2) return 0;
    // (the eight XCDs' shares of the row-blocks [lo, hi), by work)
    auto split_of = [&](size_t lo, size_t hi, uint32_t &longest) {
        XcdSplit xs;
        xs.first[0] = (uint32_t) lo;
        for (uint32_t x = 1; x < 8; ++x) {
            const uint64_t want = m->rb_upto[lo] + (m->rb_upto[hi] - m->rb_upto[lo]) * x / 8;
            size_t i = (size_t)(std::lower_bound(m->rb_upto.begin() + lo, m->rb_upto.begin() + hi + 1, want) - m->rb_upto.begin());
            xs.first[x] = (uint32_t) std::min(std::max<size_t>(i, xs.first[x - 1]), hi);
        }
        xs.first[8] = (uint32_t) hi;
        longest = 0;
        for (uint32_t x = 0; x < 8; ++x) longest = std::max(longest, xs.first[x + 1] - xs.first[x]);
        return xs;
    };
    if (base > 0) cp.front = split_of(0, base, cp.front_longest);
    std::vector<size_t> cut(K + 1, base);
    for (size_t k = 1; k < K; ++k) {
        const uint64_t want = m->rb_upto[base] + (m->rb_upto[n] - m->rb_upto[base]) * k / K;
        size_t i = (size_t)(std::lower_bound(m->rb_upto.begin(), m->rb_upto.end(), want) - m->rb_upto.begin());
        cut[k] = std::min(std::max(i, cut[k - 1]), n);
    }
    cut[K] = n;
    for (size_t k = 0; k < K; ++k) {
        const size_t lo = cut[k], hi = cut[k + 1];
        uint32_t longest = 0;
        cp.split.push_back(split_of(lo, hi, longest));
        cp.longest.push_back(longest);
        row_bounds.push_back(lo < n ? (size_t) m->rb_row0[lo] : m->own_hi);
    }
    if (m->symmetric)
        for (size_t &b : row_bounds) b = std::max(b, m->private_rows[0].first);
    if (m->symmetric) {
        row_bounds[0] = m->private_rows[0].first;
        row_bounds.push_back(m->private_rows[0].second);
    } else {
        row_bounds[0] = m->own_lo;
        row_bounds.push_back(m->own_hi);
    }
    cp.bounds = row_bounds;
    cp.order.clear();
    cp.step_pieces.clear();
    if (slot == 2 && m->rb_xneed.size() == n && m->xneed_piece) {
        const size_t P = (m->ncols + m->xneed_piece - 1) / m->xneed_piece;
        std::vector<uint64_t> need(K, 0ull);
        for (size_t k = 0; k < K; ++k)
            for (size_t i = cut[k]; i < cut[k + 1]; ++i) need[k] |= m->rb_xneed[i];
        uint64_t init_need = 0, front_need = 0;
        if (m->symmetric) {
            // the init pass in front of the part that runs first reads x of every row that does not store itself;
            // the row-blocks in front of the parts run with the one that runs last
            auto rows = [&](size_t lo, size_t hi) {
                for (size_t pc = lo / m->xneed_piece; hi > lo && pc <= (hi - 1) / m->xneed_piece && pc < P; ++pc) init_need |= 1ull << pc;
            };
            rows(0, m->private_rows[0].first);
            rows(m->private_rows[0].second, m->nrows);
            for (size_t i = 0; i < base; ++i) front_need |= m->rb_xneed[i];
        }
        uint64_t have = 0;
        std::vector<char> done(K, 0);
        for (size_t step = 0; step < K; ++step) {
            const uint64_t with = (step == 0 ? init_need : 0ull) | (step + 1 == K ? front_need : 0ull);
            size_t best = K;
            int best_new = 65;
            for (size_t k = 0; k < K; ++k) {
                const int fresh = __builtin_popcountll((need[k] | with) & ~have);
                if (!done[k] && fresh < best_new) { best_new = fresh; best = k; }
            }
            need[best] |= with;
            done[best] = 1;
            cp.order.push_back((uint32_t) best);
            std::vector<uint32_t> pcs;
            for (size_t pc = 0; pc < P; ++pc)
                if (((need[best] & ~have) >> pc) & 1ull) pcs.push_back((uint32_t) pc);
            have |= need[best];
            cp.step_pieces.push_back(pcs);
        }
        for (size_t pc = 0; pc < P; ++pc)            // (pieces nobody reads: with the last step, d_x is x as a whole)
            if (!((have >> pc) & 1ull)) cp.step_pieces.back().push_back((uint32_t) pc);
        std::string txt;
        for (size_t j = 0; j < K; ++j) {
            txt += (j ? ", " : "") + std::to_string(cp.order[j]) + " (";
            for (size_t q = 0; q < cp.step_pieces[j].size(); ++q) txt += (q ? " " : "") + std::to_string(cp.step_pieces[j][q]);
            txt += ")";
        }
        log_msg(LOG_INFO, "host vectors: %zu parts, x in %zu pieces of %.1f MB; part (pieces sent in front of it): %s\n", K, P,
                (double) m->xneed_piece * sizeof(double) / 1048576.0, txt.c_str());
    }
    return K;
}

void device_spmv_chunk(DeviceMatrix *m, size_t k, double alpha, const double *d_x, double beta, double *d_y, void *stream_, int slot,
                       int position)
{
    const DeviceMatrix::ChunkPlan &cp = m->chunks[slot < 0 || slot > 2 ? 0 : slot];
    if (k >= cp.split.size()) throw FatalError("no such part of the stream (device_plan_chunks)");
    m->launched_since_edit = true;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const XcdSplit xs = cp.split[k];
    const uint32_t blocks = 8u * cp.longest[k];
    if (m->symmetric) {
        // (the init pass goes in front of the part that is launched first, the row-blocks in front of the parts and
        // whatever follows the product behind the one that is launched last: by number unless the caller says)
        const bool first = position < 0 ? k == 0 : (position & 1) != 0;
        const bool last = position < 0 ? k + 1 == cp.split.size() : (position & 2) != 0;
        if (last && cp.front_longest) {
            SpmvPart front{cp.front, cp.front_longest, false, false};
            device_spmv_impl(m, alpha, d_x, beta, d_y, stream_, &front);
        }
        SpmvPart part{xs, cp.longest[k], first, last};
        device_spmv_impl(m, alpha, d_x, beta, d_y, stream_, &part);
        return;
    }
    if (!blocks) return;
    // (the launch tuner may have settled on a y tile per wavefront: the same parts through that kernel)
    const size_t lds = (m->wave_tiles ? (size_t) m->waves : 1u) * m->lds_doubles * sizeof(double);
#define SPX_LAUNCH_CHUNK_K(KERNEL, W)                                                                        \
    hipLaunchKernelGGL(KERNEL<W>, dim3(blocks), dim3(64 * W), lds, stream, m->rbs, m->passes, m->n_rb,       \
                       m->pass_stride, xs, m->values, m->descs, m->cidx, m->segrows, d_x, d_y, m->carry,        \
                       (const double *) nullptr, (double *) nullptr, (const uint32_t *) nullptr, alpha, beta,  \
                       (const double *) nullptr, 0.0)
#define SPX_LAUNCH_CHUNK(W)                                                                                  \
    do {                                                                                                     \
        if (m->wave_tiles) SPX_LAUNCH_CHUNK_K(csx_spmv_det_kernel, W);                                       \
        else SPX_LAUNCH_CHUNK_K(csx_spmv_kernel, W);                                                         \
    } while (0)
    if (m->xw_on && m->passes_xw && !m->wave_tiles) {
        KernelArgs ax;
        memset(&ax, 0, sizeof(ax));
        ax.rbs = m->rbs; ax.passes = m->passes_xw; ax.values = m->values; ax.descs = m->xdescs;
        ax.cidx = m->cidx; ax.segrows = m->segrows; ax.x = d_x; ax.y = d_y; ax.carry = m->carry;
        ax.alpha = alpha; ax.beta = beta; ax.n_rb = m->n_rb; ax.pass_stride = m->pass_stride;
        ax.xw_tab = m->xw_tab;
        launch_spmv_xw(m->waves, blocks, (size_t) m->lds_doubles_xw * sizeof(double), stream, ax, xs);
    }
    else if (m->waves == 2) SPX_LAUNCH_CHUNK(2);
    else if (m->waves == 8) SPX_LAUNCH_CHUNK(8);
    else SPX_LAUNCH_CHUNK(4);
#undef SPX_LAUNCH_CHUNK_K
#undef SPX_LAUNCH_CHUNK
    HIP_CHECK(hipGetLastError());
}

void device_set_init_rows(DeviceMatrix *m, size_t first_row)
{
    m->init_lo = first_row;
    m->init_limited = true;
}

void device_set_sym_atomic(DeviceMatrix *m, bool on)
{
    m->sym_atomic = (on || m->has_symsegs) && m->has_tiles && !m->wave_tiles;
}

// per-wavefront tiles need waves x the LDS: pick the largest wavefront count that fits
// (the kernels may use up to 160 KB once told so)
void device_set_wave_tiles(DeviceMatrix *m, bool on)
{
    m->wave_tiles = on;
    if (!on) {
        m->waves = m->waves_req;       // (a trial with per-wavefront tiles may have lowered it)
        return;
    }
    m->sym_atomic = false;
    const size_t per_copy = (size_t) m->lds_doubles * sizeof(double);
    int w = m->waves_req;
    while (w > 2 && (size_t) w * per_copy > 160u * 1024u) w /= 2;
    if ((size_t) w * per_copy > 160u * 1024u) throw FatalError("row-blocks too large for per-wavefront tiles");
    m->waves = w;
    const int bytes = 160 * 1024;
#define SPX_ATTR(K) (void) hipFuncSetAttribute(reinterpret_cast<const void *>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, bytes)
    SPX_ATTR(csx_spmv_det_kernel<2>); SPX_ATTR(csx_spmv_det_kernel<4>); SPX_ATTR(csx_spmv_det_kernel<8>);
    SPX_ATTR(csx_spmv_symtile_det_kernel<2>); SPX_ATTR(csx_spmv_symtile_det_kernel<4>);
    SPX_ATTR(csx_spmv_symtile_det_kernel<8>);
#undef SPX_ATTR
}
bool device_get_wave_tiles(const DeviceMatrix *m) { return m->wave_tiles; }
bool device_has_tiles(const DeviceMatrix *m) { return m->has_tiles; }

void device_set_deterministic(DeviceMatrix *m, bool on)
{
    m->deterministic = on;
    if (on) device_set_wave_tiles(m, true);
}
bool device_get_deterministic(const DeviceMatrix *m) { return m->deterministic; }
bool device_get_sym_atomic(const DeviceMatrix *m) { return m->sym_atomic; }
bool device_has_spill(const DeviceMatrix *m) { return m->has_tiles && m->n_spill; }

void device_set_waves(DeviceMatrix *m, int waves)
{
    m->waves = m->waves_req = (waves == 2 || waves == 8) ? waves : 4;
    if (m->wave_tiles) device_set_wave_tiles(m, true);     // (re-checks the LDS budget)
}

int device_get_waves(const DeviceMatrix *m) { return m->waves; }

int device_host_parts(const DeviceMatrix *m) { return m ? m->host_parts : 0; }
void device_set_host_parts(DeviceMatrix *m, size_t parts) { m->host_parts_want = std::min<size_t>(parts, 64); }

int device_host_order(const DeviceMatrix *m, int32_t *order, int cap)
{
    if (!m || m->host_parts < 2 || !m->host_x_by_need) return 0;
    const DeviceMatrix::ChunkPlan &cp = m->chunks[2];
    for (int j = 0; j < cap && j < (int) cp.order.size(); ++j) order[j] = (int32_t) cp.order[(size_t) j];
    return (int) cp.order.size();
}

bool device_has_sx(const DeviceMatrix *m) { return m->passes_sx != nullptr; }
void device_set_sx(DeviceMatrix *m, bool on) { m->sx_on = on && m->passes_sx && m->sym_atomic && !m->wave_tiles; }
bool device_get_sx(const DeviceMatrix *m) { return m->sx_on && m->passes_sx && m->sym_atomic && !m->wave_tiles; }
void device_sx_info(const DeviceMatrix *m, uint64_t &elems_sx, uint64_t &elems_sym, size_t &rowblocks)
{
    elems_sx = m->sx_elems;
    elems_sym = m->sx_sym_elems;
    rowblocks = m->sx_rowblocks;
}
bool device_has_xw(const DeviceMatrix *m) { return m->passes_xw != nullptr; }
void device_set_xw(DeviceMatrix *m, bool on) { m->xw_on = on && m->passes_xw && !m->wave_tiles; }
bool device_get_xw(const DeviceMatrix *m) { return m->xw_on && m->passes_xw && !m->wave_tiles; }
void device_xw_info(const DeviceMatrix *m, uint64_t &elems_lds, uint64_t &unit_elems, uint64_t &staged, uint32_t &lds_bytes)
{
    elems_lds = m->xw_elems;
    unit_elems = m->xw_unit_elems;
    staged = m->xw_staged;
    lds_bytes = (uint32_t) (m->lds_doubles_xw * sizeof(double));
}

static void ensure_staging(DeviceMatrix *m)
{
    const size_t xb = m->ncols * sizeof(double), yb = m->nrows * sizeof(double);
    if (!m->d_x) {
        HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&m->d_x), xb ? xb : 8));
        HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&m->d_y), yb ? yb : 8));
    }
    if (!m->p_x) {
        HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&m->p_x), xb ? xb : 8, hipHostMallocDefault));
        HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&m->p_y), yb ? yb : 8, hipHostMallocDefault));
    }
    if (!m->host_stream) HIP_CHECK(hipStreamCreateWithFlags(&m->host_stream, hipStreamNonBlocking));
}

double device_time_spmv(DeviceMatrix *m, int warmup, int launches)
{
    HIP_CHECK(hipSetDevice(m->device));
    // (scratch vectors in HBM and a stream only: the pinned host buffers of the host-vector entry
    // point -- 2 x 224 MB on the contract matrix -- are allocated when that entry point is first used)
    if (!m->d_x) {
        const size_t xb = m->ncols * sizeof(double), yb = m->nrows * sizeof(double);
        HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&m->d_x), xb ? xb : 8));
        HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&m->d_y), yb ? yb : 8));
    }
    if (!m->host_stream) HIP_CHECK(hipStreamCreateWithFlags(&m->host_stream, hipStreamNonBlocking));
    hipStream_t st = m->host_stream;
    HIP_CHECK(hipMemsetAsync(m->d_x, 0, m->ncols * sizeof(double), st));
    HIP_CHECK(hipMemsetAsync(m->d_y, 0, m->nrows * sizeof(double), st));
    m->x_version = 0;
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    for (int i = 0; i < warmup; ++i) device_spmv(m, 1.0, m->d_x, 0.0, m->d_y, st);
    HIP_CHECK(hipEventRecord(e0, st));
    for (int i = 0; i < launches; ++i) device_spmv(m, 1.0, m->d_x, 0.0, m->d_y, st);
    HIP_CHECK(hipEventRecord(e1, st));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void) hipEventDestroy(e0);
    (void) hipEventDestroy(e1);
    return 1e-3 * ms / (launches > 0 ? launches : 1);
}

// Host-vector entry point: x (and y when it is read) go through pinned staging
// buffers and asynchronous copies on one private stream -- a pageable
// hipMemcpy stages internally as well, but synchronously and chunk by chunk.
// Large user buffers are staged in pieces, a few threads copying a piece while the
// DMA engine moves the one in front of it (and the other way round on the way back):
// the reference's clients hand over plain malloc'ed vectors (SPX_VEC_AS_IS), and a
// single-threaded copy of 224 MB into the staging buffer took longer than the DMA.
namespace {


void copy_threads(void *dst, const void *src, size_t bytes)
{
    const unsigned t = bytes >= ((size_t) 4 << 20) ? std::min(4u, host_threads()) : 1u;
    if (t <= 1) {
        std::memcpy(dst, src, bytes);
        return;
    }
    const size_t part = ((bytes / t) + 63) & ~(size_t) 63;
    parallel_for(t, t, [&](size_t k) {
        const size_t a = std::min(bytes, k * part), b = k + 1 == t ? bytes : std::min(bytes, (k + 1) * part);
        if (b > a) std::memcpy(static_cast<char *>(dst) + a, static_cast<const char *>(src) + a, b - a);
    });
}

// host (pageable) -> device through the pinned buffer `stage`
void upload_staged(double *d, double *stage, const double *h, size_t bytes, hipStream_t st)
{
    for (size_t off = 0; off < bytes; off += STAGE_PIECE) {
        const size_t n = std::min(STAGE_PIECE, bytes - off);
        copy_threads(reinterpret_cast<char *>(stage) + off, reinterpret_cast<const char *>(h) + off, n);
        HIP_CHECK(hipMemcpyAsync(reinterpret_cast<char *>(d) + off, reinterpret_cast<char *>(stage) + off, n,
                                 hipMemcpyHostToDevice, st));
    }
}

}  // namespace

void device_spmv_host(DeviceMatrix *m, double alpha, const double *h_x, bool x_pinned,
                      double beta, double *h_y, bool y_pinned,
                      const std::function<void(double *, void *)> &after, uint64_t x_version)
{
    HIP_CHECK(hipSetDevice(m->device));
    const size_t xb = m->ncols * sizeof(double), yb = m->nrows * sizeof(double);
    ensure_staging(m);
    hipStream_t st = m->host_stream;
    // A large y goes back in parts behind the product: the stream is cut into parts of whole rows (where it can be:
    // device_plan_chunks), every part's rows start on their way as soon as its kernel has ended, on a stream of
    // their own, while the next part runs -- the download of the bench matrix's 224 MB takes three times as long as
    // its product.  And x comes in pieces in the order the parts need them (general streams: the plan's `order`),
    // so that the two directions of the link are busy at the same time.
    const bool whole = m->own_lo == 0 && m->own_hi == m->nrows && (!m->symmetric || m->sym_fused);
    std::vector<size_t> bounds;
    // (by need: page-locked x only -- a pageable x goes through staging memory in large pieces, where the host's
    // copying, not the link, sets the pace)
    const bool send_x = !x_version || x_version != m->x_version;
    const bool want_by_need = send_x && x_pinned && !m->rb_xneed.empty();
    const size_t K = (!after && whole && yb >= host_parts_min_bytes())
                         ? device_plan_chunks(m, m->host_parts_want ? m->host_parts_want : (want_by_need ? (m->symmetric ? HOST_PARTS_X_SYM : HOST_PARTS_X) : HOST_PARTS), bounds, 2) : 0;
    const DeviceMatrix::ChunkPlan &cp = m->chunks[2];
    const bool x_by_need = want_by_need && K >= 2 && cp.order.size() == K && cp.step_pieces.size() == K;
    if (send_x && !x_by_need) {
        if (x_pinned) HIP_CHECK(hipMemcpyAsync(m->d_x, h_x, xb, hipMemcpyHostToDevice, st));
        else upload_staged(m->d_x, m->p_x, h_x, xb, st);
        m->x_version = x_version;
    }
    // y travels to the device only when it is read: beta != 0, or this process
    // owns a slice of the rows and the others must keep the caller's values
    // (atomic hand-over reads y only through the init kernel's beta*y: nothing to upload when beta == 0)
    // (... and where x goes up by need, a page-locked y that is read goes up the same way: a part reads its own
    // rows, the init pass of a symmetric stream the rows that do not store themselves)
    const bool y_by_need = x_by_need && y_pinned && beta != 0.0;
    if ((beta != 0.0 || !whole) && !y_by_need) {
        if (y_pinned) HIP_CHECK(hipMemcpyAsync(m->d_y, h_y, yb, hipMemcpyHostToDevice, st));
        else upload_staged(m->d_y, m->p_y, h_y, yb, st);
    }
    m->host_parts = (int) K;
    m->host_x_by_need = x_by_need;
    if (K >= 2) {
        if (!m->copy_stream) HIP_CHECK(hipStreamCreateWithFlags(&m->copy_stream, hipStreamNonBlocking));
        if (x_by_need && !m->up_stream) HIP_CHECK(hipStreamCreateWithFlags(&m->up_stream, hipStreamNonBlocking));
        // the pieces of y and the part each of them is final behind: a part's own rows; and (symmetric streams,
        // whose parts cover the rows that store themselves only) whatever lies outside them behind the LAST part
        struct Piece { size_t lo, hi; };
        std::vector<Piece> pieces;
        std::vector<std::vector<size_t>> pieces_of(K);
        auto add_piece = [&](size_t lo, size_t hi, size_t part) {
            pieces_of[part].push_back(pieces.size());
            pieces.push_back(Piece{lo, hi});
        };
        for (size_t k = 0; k < K; ++k)
            if (bounds[k + 1] > bounds[k]) add_piece(bounds[k], bounds[k + 1], k);
        const size_t k_last = cp.order.size() == K ? cp.order[K - 1] : K - 1;       // (the part that is launched last)
        const size_t n_own = pieces.size();                                          // (pieces [n_own, ..): outside the parts)
        if (bounds[0] > 0) add_piece(0, bounds[0], k_last);
        if (bounds[K] < m->nrows) add_piece(bounds[K], m->nrows, k_last);
        // events: [0, K) behind the parts, then one behind every piece of y, then [.., + K) behind the steps of x
        const size_t ev_piece = K, ev_up = K + pieces.size();
        while (m->part_events.size() < ev_up + K) {
            hipEvent_t e;
            HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            m->part_events.push_back(e);
        }
        // (a failure in here must not leave copies queued that still write into the caller's y -- or read the
        // caller's x -- after the C entry point has returned its error: the streams are drained before the
        // exception travels on)
        try {
            if (x_by_need) {
                // (whatever the product's stream was given before -- y on its way up -- is not waited for: x only)
                m->x_version = 0;
            }
            for (size_t j = 0; j < K; ++j) {
                const size_t k = cp.order.size() == K ? cp.order[j] : j;
                if (x_by_need) {
                    for (uint32_t pc : cp.step_pieces[j]) {
                        const size_t off = (size_t) pc * m->xneed_piece * sizeof(double);
                        if (off >= xb) continue;
                        const size_t n = std::min(m->xneed_piece * sizeof(double), xb - off);
                        HIP_CHECK(hipMemcpyAsync(reinterpret_cast<char *>(m->d_x) + off, reinterpret_cast<const char *>(h_x) + off, n,
                                                 hipMemcpyHostToDevice, m->up_stream));
                    }
                    if (y_by_need) {
                        auto send_y = [&](const Piece &pc) {
                            const size_t off = pc.lo * sizeof(double), n = (pc.hi - pc.lo) * sizeof(double);
                            HIP_CHECK(hipMemcpyAsync(reinterpret_cast<char *>(m->d_y) + off, reinterpret_cast<const char *>(h_y) + off, n,
                                                     hipMemcpyHostToDevice, m->up_stream));
                        };
                        if (j == 0)
                            for (size_t i = n_own; i < pieces.size(); ++i) send_y(pieces[i]);
                        for (size_t i : pieces_of[k])
                            if (i < n_own) send_y(pieces[i]);
                    }
                    HIP_CHECK(hipEventRecord(m->part_events[ev_up + j], m->up_stream));
                    HIP_CHECK(hipStreamWaitEvent(st, m->part_events[ev_up + j], 0));
                }
                device_spmv_chunk(m, k, alpha, m->d_x, beta, m->d_y, st, 2, (j == 0 ? 1 : 0) | (j + 1 == K ? 2 : 0));
                HIP_CHECK(hipEventRecord(m->part_events[k], st));
                HIP_CHECK(hipStreamWaitEvent(m->copy_stream, m->part_events[k], 0));
                for (size_t i : pieces_of[k]) {
                    const size_t off = pieces[i].lo * sizeof(double), n = (pieces[i].hi - pieces[i].lo) * sizeof(double);
                    char *dst = reinterpret_cast<char *>(y_pinned ? h_y : m->p_y) + off;
                    HIP_CHECK(hipMemcpyAsync(dst, reinterpret_cast<char *>(m->d_y) + off, n, hipMemcpyDeviceToHost, m->copy_stream));
                    if (!y_pinned) HIP_CHECK(hipEventRecord(m->part_events[ev_piece + i], m->copy_stream));
                }
            }
            if (y_pinned) {
                HIP_CHECK(hipStreamSynchronize(m->copy_stream));
            } else {
                // (in the order they were sent)
                for (size_t j = 0; j < K; ++j)
                    for (size_t i : pieces_of[cp.order.size() == K ? cp.order[j] : j]) {
                        const size_t off = pieces[i].lo * sizeof(double), n = (pieces[i].hi - pieces[i].lo) * sizeof(double);
                        HIP_CHECK(hipEventSynchronize(m->part_events[ev_piece + i]));
                        copy_threads(reinterpret_cast<char *>(h_y) + off, reinterpret_cast<char *>(m->p_y) + off, n);
                    }
            }
            // (everything that was enqueued has run by now -- the last rows of y were behind the last part, which was
            // behind the last piece of x; the two waits cost some microseconds and make sure of it whatever the cut:
            // the caller may release its vectors the moment this returns)
            if (m->up_stream) HIP_CHECK(hipStreamSynchronize(m->up_stream));
            HIP_CHECK(hipStreamSynchronize(st));
            if (x_by_need) m->x_version = x_version;
            return;
        } catch (...) {
            if (m->up_stream) (void) hipStreamSynchronize(m->up_stream);
            (void) hipStreamSynchronize(m->copy_stream);
            (void) hipStreamSynchronize(st);
            (void) hipGetLastError();
            m->host_parts = 0;
            m->host_x_by_need = false;
            if (x_by_need) m->x_version = 0;
            throw;
        }
    }
    m->host_parts = 0;
    device_spmv(m, alpha, m->d_x, beta, m->d_y, st);
    if (after) after(m->d_y, st);
    if (y_pinned) {
        HIP_CHECK(hipMemcpyAsync(h_y, m->d_y, yb, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        return;
    }
    // back in pieces: an event behind every piece, the host copies a piece out of the
    // staging buffer while the next one is on its way
    const size_t pieces = (yb + STAGE_PIECE - 1) / STAGE_PIECE;
    if (pieces <= 1) {
        HIP_CHECK(hipMemcpyAsync(m->p_y, m->d_y, yb, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        std::memcpy(h_y, m->p_y, yb);
        return;
    }
    while (m->stage_events.size() < pieces) {
        hipEvent_t e;
        HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        m->stage_events.push_back(e);
    }
    for (size_t k = 0; k < pieces; ++k) {
        const size_t off = k * STAGE_PIECE, n = std::min(STAGE_PIECE, yb - off);
        HIP_CHECK(hipMemcpyAsync(reinterpret_cast<char *>(m->p_y) + off, reinterpret_cast<char *>(m->d_y) + off, n,
                                 hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipEventRecord(m->stage_events[k], st));
    }
    for (size_t k = 0; k < pieces; ++k) {
        const size_t off = k * STAGE_PIECE, n = std::min(STAGE_PIECE, yb - off);
        HIP_CHECK(hipEventSynchronize(m->stage_events[k]));
        copy_threads(reinterpret_cast<char *>(h_y) + off, reinterpret_cast<char *>(m->p_y) + off, n);
    }
}

bool device_stream_is_capturing(void *stream)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(static_cast<hipStream_t>(stream), &st) != hipSuccess) {
        (void) hipGetLastError();
        return false;
    }
    return st != hipStreamCaptureStatusNone;
}

void *device_host_alloc(size_t bytes)
{
    if (device_count() <= 0) return nullptr;
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 8, hipHostMallocPortable) != hipSuccess) {
        (void) hipGetLastError();
        return nullptr;
    }
    return p;
}

void device_host_free(void *p)
{
    if (p) (void) hipHostFree(p);
}

size_t device_host_parts_min_bytes() { return host_parts_min_bytes(); }

int device_host_register(void *p, size_t bytes)
{
    if (!p || !bytes || device_count() <= 0) return 0;
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    if (e == hipSuccess) return 1;
    (void) hipGetLastError();
    return e == hipErrorHostMemoryAlreadyRegistered ? 2 : 0;
}

void device_host_unregister(void *p)
{
    if (!p) return;
    if (hipHostUnregister(p) != hipSuccess) (void) hipGetLastError();
}

template <typename T, typename A>
static void download(std::vector<T, A> &v, const T *d, size_t n)
{
    v.resize(n);
    if (n) HIP_CHECK(hipMemcpy(v.data(), d, n * sizeof(T), hipMemcpyDeviceToHost));
}

void device_download(const DeviceMatrix *m, GpuStream &s)
{
    HIP_CHECK(hipSetDevice(m->device));
    download(s.rbs, m->rbs, m->n_rb);
    download(s.values, m->values, m->n_values);
    download(s.descs, m->descs, m->n_descs);
    download(s.passes, m->passes, m->n_passes);
    if (!m->launch_order.empty()) {
        // back into the stream's own (ascending) order
        const size_t stride = m->pass_stride;
        std::vector<SpxRowBlock> rbs(s.rbs.size());
        std::vector<SpxPass> passes(s.passes.size());
        for (size_t i = 0; i < m->launch_order.size(); ++i) {
            const size_t o = m->launch_order[i];
            rbs[o] = s.rbs[i];
            rbs[o].pass_off = (uint32_t)(o * stride);
            std::copy(s.passes.begin() + i * stride, s.passes.begin() + (i + 1) * stride, passes.begin() + o * stride);
        }
        s.rbs.swap(rbs);
        s.passes.swap(passes);
    }
    download(s.cidx, m->cidx, m->n_cidx);
    download(s.segrows, m->segrows, m->n_segrows);
    download(s.shared, m->shared, m->n_shared);
    s.n_carry = m->n_carry;
    s.sym_fused = m->sym_fused;
    s.pass_stride = m->pass_stride;
    s.waves = (uint32_t) m->waves;
    s.n_spill = m->n_spill;
    s.lds_doubles = m->lds_doubles;
    if (m->n_mirror_rows) {
        download(s.mirror_rows, m->mirror_rows, m->n_mirror_rows);
        download(s.mirror_ptr, m->mirror_ptr, (size_t) m->n_mirror_rows + 1);
        download(s.mirror_col, m->mirror_col, m->n_mirror_nnz);
        download(s.mirror_val, m->mirror_val, m->n_mirror_nnz);
    }
    s.sym_atomic = m->sym_atomic;
    s.deterministic = m->deterministic;
    s.wave_tiles = m->wave_tiles;
    s.xw_on = device_get_xw(m);
    s.sx_plan = m->passes_sx != nullptr;
    s.sx_on = device_get_sx(m);
    s.xw_budget = m->xw_budget;
    s.xw_gap = m->xw_gap;
    if (m->n_spill) download(s.slot_group_col, m->slot_col, m->n_slot_col);
    if (m->n_spill) {
        download(s.fix_ptr, m->fix_ptr, m->n_fix_ptr);
        download(s.fix_idx, m->fix_idx, m->n_fix_idx);
    }
    if (m->symmetric) download(s.dvalues, m->dvalues, m->nrows);
}

// Products enqueued on a non-blocking stream (spx_hip_matvec_*) are not ordered against a blocking
// copy by themselves: before the first value changes after a product was enqueued, wait for whatever
// the device still runs -- ONCE, not per entry (a client that refreshes every value through
// spx_mat_set_entry would pay a device-wide wait per nonzero).  A stream of the device that is being
// captured makes the wait fail: that is reported, the value is not touched.
static void quiesce_before_edit(DeviceMatrix *m)
{
    if (!m->launched_since_edit) return;
    const hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void) hipGetLastError();
        throw FatalError(std::string("cannot change a value while the device cannot be waited for (") + hipGetErrorString(e) +
                         "): is a stream being captured?");
    }
    m->launched_since_edit = false;
}

void device_poke_mirror(DeviceMatrix *m, size_t index, double value)
{
    if (index >= m->n_mirror_nnz) throw FatalError("value index outside the mirror list");
    HIP_CHECK(hipSetDevice(m->device));
    quiesce_before_edit(m);
    HIP_CHECK(hipMemcpy(m->mirror_val + index, &value, sizeof(value), hipMemcpyHostToDevice));
}

double device_peek(const DeviceMatrix *m, bool diagonal, size_t index)
{
    if (diagonal ? (!m->dvalues || index >= m->nrows) : index >= m->n_values)
        throw FatalError("value index outside the stream");
    HIP_CHECK(hipSetDevice(m->device));
    double v = 0.0;
    HIP_CHECK(hipMemcpy(&v, (diagonal ? m->dvalues : m->values) + index, sizeof(v), hipMemcpyDeviceToHost));
    return v;
}

void device_poke(DeviceMatrix *m, bool diagonal, size_t index, double value)
{
    if (diagonal ? (!m->dvalues || index >= m->nrows) : index >= m->n_values)
        throw FatalError("value index outside the stream");
    HIP_CHECK(hipSetDevice(m->device));
    quiesce_before_edit(m);
    HIP_CHECK(hipMemcpy((diagonal ? m->dvalues : m->values) + index, &value, sizeof(value),
                        hipMemcpyHostToDevice));
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
