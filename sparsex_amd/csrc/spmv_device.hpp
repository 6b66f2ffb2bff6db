// spmv_device.hpp -- device code shared by the kernels of the CSX interpreter (spmv_kernels.hip: the
// general and the symmetric kernels; spmv_xw_kernels.hip: the general kernel with the unit windows of x
// in LDS): kernel arguments, the XCD-aware row-block order, and the pass bodies -- one lane per row
// segment, values interleaved, x gathered through L2 or read from the row-block's LDS window.
//
// Semantics restated from the reference's SpMV templates (src/templates/csx_spmv_tmpl.c:66-101 and the
// per-unit bodies delta/horiz/vert/diag/rdiag/block_row/block_col _tmpl.c).
#pragma once

#include "gpu_format.h"
#include "xwindows.hpp"

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace spx {

struct KernelArgs {
    const SpxRowBlock *rbs;
    const SpxPass *passes;
    const double *values;
    const SpxUnitDesc *descs;
    const uint8_t *cidx;
    const uint16_t *segrows;
    const double *x;
    double *y;
    double *carry;
    const double *dvalues;   // symmetric, fused: diagonal added at the write-out (else null)
    double *spill;           // symmetric tiles: transposed sums of columns owned by other row-blocks
    const uint32_t *slot_col;  // ... or (atomic hand-over) the first column of every group of eight slots
    double alpha, beta;
    const double *dvalues_priv;   // atomic hand-over: diagonal for the row-blocks that store their rows
    double beta_priv;             // ... and the caller's beta for them (beta above is 1 after the init pass)
    uint32_t n_rb;
    uint32_t pass_stride;    // pass headers of row-block i start at passes[i * pass_stride]
    const XwEntry *xw_tab;   // unit windows of x (xwindows.hpp): XW_MAX entries per row-block, or null
};

// XCD-aware order of the row-blocks: workgroup b runs on XCD b % 8; XCD x walks the row-blocks
// [first[x], first[x + 1]) in turn, a contiguous part of the matrix that holds an eighth of its
// VALUES (not of its row-blocks: a symmetric KKT matrix keeps its stored triangle in the second
// half of its rows, and an eighth of the row-blocks by count left five XCDs without work)
struct XcdSplit {
    uint32_t first[9];
};

// wavefronts per workgroup: the kernels exist for 2, 4 and 8 (spx.gpu.waves, or
// measured at tune time: small matrices like 2, leftover-heavy ones 8)
constexpr int MAX_WAVES_PER_BLOCK = 8;

// Loads of the matrix stream (values, descriptors): plain loads.  (Marking them non-temporal, so that
// they would not push x out of the L2, measured slower on every workload: profiles/r03/ablation.md
// section 4, profiles/r05/xw_nontemporal_raw.md; the experiment hook is gone, the record stays.)
typedef double spx_d2_t __attribute__((ext_vector_type(2)));
// two doubles at any 8-byte aligned address as ONE load (global_load_dwordx4 needs no 16-byte
// alignment on gfx9): the x of a row segment comes in pairs wherever its first column lies
typedef double spx_d2u_t __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ double2 ld_stream(const double2 *p) { return *p; }
__device__ __forceinline__ double ld_stream(const double *p) { return *p; }
__device__ __forceinline__ uint2 ld_stream(const uint2 *p) { return *p; }
#define SPX_LD_INDEX(expr) (expr)

// set bits of `mask` in lanes 1..lane (bit 0 is never set by the emitter)
__device__ __forceinline__ uint32_t starts_upto(uint64_t mask, int lane)
{
    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                     __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
    return below + (uint32_t)((mask >> lane) & 1ull);
}

// B unit passes of the same width at once: lane l owns one row segment of W
// consecutive columns in each of them.  All descriptor loads go out first,
// then all value loads, then the x gathers: one memory round trip per stage
// for the whole batch instead of one per pass.
//
// G (gather pass): the lane's segment is a piece of one row's leftover
// nonzeros; its row comes from the row-block's u16 rows and every nonzero has
// its own column offset (element-major [W][nseg]) instead of a descriptor.
//
// G == 2 (SPX_PASS_GATHER_LDS): the same, but the columns lie in the row-block's
// x window, which the workgroup has staged in LDS (`win`): u16 offsets, ds_read.
template <int W, int B, int G>
__device__ __forceinline__ void unit_passes(const KernelArgs &a, const SpxRowBlock &rb,
                                            const SpxPass (&ps)[B], double *tile,
                                            const double *win, int lane)
{
    bool active[B];
    uint32_t l[B], nseg[B];
    uint2 q[B];
    uint32_t goff[B][G ? W : 1];
#pragma unroll
    for (int b = 0; b < B; ++b) {
        nseg[b] = ps[b].nseg;
        active[b] = (uint32_t) lane < nseg[b];
        l[b] = active[b] ? (uint32_t) lane : 0u;         // idle lanes shadow lane 0
        if (G) {
            q[b].x = SPX_LD_INDEX(a.segrows[rb.seg_off + ps[b].seg0 + l[b]]);
            const uint8_t *cidx = a.cidx + ((size_t) rb.cidx_off + (G == 2 ? rb.near_off : 0u)) * 16u;
            const uint32_t e0 = ps[b].elem0 + l[b];
            if (G == 1 && rb.cidx_width == 4) {
#pragma unroll
                for (int w = 0; w < W; ++w)
                    goff[b][w] = SPX_LD_INDEX(reinterpret_cast<const uint32_t *>(cidx)[e0 + (uint32_t) w * nseg[b]]);
            } else if (G == 1 && rb.cidx_width == 3) {
                // 24-bit offsets: the low halves, then (array of its own) the high bytes
                const uint8_t *hi = cidx + (size_t) rb.hi_off * 16u;
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    const uint32_t e = e0 + (uint32_t) w * nseg[b];
                    goff[b][w] = (uint32_t) SPX_LD_INDEX(reinterpret_cast<const uint16_t *>(cidx)[e]) | ((uint32_t) SPX_LD_INDEX(hi[e]) << 16);
                }
            } else {
#pragma unroll
                for (int w = 0; w < W; ++w)
                    goff[b][w] = SPX_LD_INDEX(reinterpret_cast<const uint16_t *>(cidx)[e0 + (uint32_t) w * nseg[b]]);
            }
        } else {
            if (ps[b].flags & SPX_PASSF_INLINE) {
                // the pass' only descriptor came with its header (wave-uniform, in SGPRs)
                q[b].x = (uint32_t) ps[b].mask;
                q[b].y = (uint32_t) (ps[b].mask >> 32);
            } else
            {
                const uint64_t mk = (ps[b].flags & SPX_PASSF_INLINE) ? 0ull : ps[b].mask;
                const uint32_t rank = (uint32_t) ps[b].rank0 + (active[b] ? starts_upto(mk, lane) : 0u);
                q[b] = ld_stream(reinterpret_cast<const uint2 *>(a.descs + rb.desc_off + rank));
            }
        }
    }
    double2 v2[B][W / 2 > 0 ? W / 2 : 1];
    double v1[B];
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const double *vals = a.values + rb.val_off + ps[b].val_off;
#pragma unroll
        for (int p = 0; p < W / 2; ++p)
            v2[b][p] = ld_stream(reinterpret_cast<const double2 *>(vals + (uint32_t) p * 2u * nseg[b] + l[b] * 2u));
        if (W & 1) v1[b] = ld_stream(vals + (uint32_t) (W / 2) * 2u * nseg[b] + l[b]);
    }
    int row[B];
    double acc[B];
    double x[B][W];
#pragma unroll
    for (int b = 0; b < B; ++b) {
        if (G == 2) {
            // (a piece shorter than the pass is padded: nothing is multiplied there)
            row[b] = (int) SPX_SEGROW_ROW(q[b].x);
            const int len = (int) SPX_SEGROW_LEN(q[b].x);
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const double xv = win[goff[b][w]];
                x[b][w] = w < len ? xv : 0.0;
            }
        } else if (G) {
            row[b] = (int) SPX_SEGROW_ROW(q[b].x);
            const int len = (int) SPX_SEGROW_LEN(q[b].x);
            const double *xp = a.x + rb.cbase;
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const double xv = xp[goff[b][w]];
                x[b][w] = w < len ? xv : 0.0;
            }
        } else {
            // segment index inside its unit, then its row / first column
            const uint32_t bits = q[b].y;
            const int s = (int) ((ps[b].seg0 + l[b] - ((bits >> 9) & 8191u)) & 0xffffu);
            const uint32_t kind = (bits >> 22) & 7u;
            const int step = (int) (bits >> 25);
            const int drow = kind == SPX_KIND_BLOCK ? 1 : (kind >= SPX_KIND_VERT ? step : 0);
            const int dcol = (kind == SPX_KIND_HORIZ || kind == SPX_KIND_DIAG)
                                 ? step : (kind == SPX_KIND_ADIAG ? -step : 0);
            row[b] = (int) (ps[b].elem0 + (bits & 511u)) + s * drow;
            const uint32_t col = q[b].x + (uint32_t) (s * dcol);
            const double *xp = a.x + col;
            // The x loads cost address-unit issue slots like the value loads do: they come in pairs
            // at any alignment, W / 2 + (W & 1) load instructions instead of W.  (One full-width load
            // per diagonal stack with the other W - 1 columns taken from the neighbouring lanes by
            // DPP shifts was built and measured slower: profiles/r03/ablation.md section 6.)
            if (W >= 2) {
                const spx_d2u_t *xp2 = reinterpret_cast<const spx_d2u_t *>(xp);
#pragma unroll
                for (int p = 0; p < W / 2; ++p) {
                    const spx_d2u_t xx = xp2[p];
                    x[b][2 * p] = xx.x;
                    x[b][2 * p + 1] = xx.y;
                }
                if (W & 1) x[b][W - 1] = xp[W - 1];
            } else {
#pragma unroll
                for (int w = 0; w < W; ++w) x[b][w] = xp[w];
            }
        }
        double t = 0.0;
#pragma unroll
        for (int p = 0; p < W / 2; ++p) {
            t = fma(v2[b][p].x, x[b][2 * p], t);
            t = fma(v2[b][p].y, x[b][2 * p + 1], t);
        }
        if (W & 1) t = fma(v1[b], x[b][W - 1], t);
        acc[b] = t;
    }
    if (G == 1 && rb.n_rows == 1) {
        // a chunk of one over-long row: every lane targets tile[0]
        double t = 0.0;
#pragma unroll
        for (int b = 0; b < B; ++b) t += active[b] ? acc[b] : 0.0;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) t += __shfl_xor(t, d);
        if (lane == 0) atomicAdd(&tile[0], t);
        return;
    }
#pragma unroll
    for (int b = 0; b < B; ++b)
        if (active[b]) atomicAdd(&tile[row[b]], acc[b]);
}

template <int B, int G>
__device__ __forceinline__ void run_units(const KernelArgs &a, const SpxRowBlock &rb,
                                          const SpxPass (&ps)[B], double *tile, const double *win,
                                          int lane)
{
    switch (ps[0].width) {         // wave-uniform
    case 1: unit_passes<1, B, G>(a, rb, ps, tile, win, lane); break;
    case 2: unit_passes<2, B, G>(a, rb, ps, tile, win, lane); break;
    case 3: unit_passes<3, B, G>(a, rb, ps, tile, win, lane); break;
    case 4: unit_passes<4, B, G>(a, rb, ps, tile, win, lane); break;
    case 5: unit_passes<5, 1, G>(a, rb, {ps[0]}, tile, win, lane);
            if (B > 1) unit_passes<5, 1, G>(a, rb, {ps[B - 1]}, tile, win, lane);
            break;
    case 6: unit_passes<6, 1, G>(a, rb, {ps[0]}, tile, win, lane);
            if (B > 1) unit_passes<6, 1, G>(a, rb, {ps[B - 1]}, tile, win, lane);
            break;
    case 7: unit_passes<7, 1, G>(a, rb, {ps[0]}, tile, win, lane);
            if (B > 1) unit_passes<7, 1, G>(a, rb, {ps[B - 1]}, tile, win, lane);
            break;
    default: unit_passes<8, 1, G>(a, rb, {ps[0]}, tile, win, lane);
            if (B > 1) unit_passes<8, 1, G>(a, rb, {ps[B - 1]}, tile, win, lane);
            break;
    }
}

__device__ __forceinline__ void run_pass(const KernelArgs &a, const SpxRowBlock &rb,
                                         const SpxPass &ps, double *tile, const double *win, int lane)
{
    if (ps.kind == SPX_PASS_GATHER) run_units<1, 1>(a, rb, {ps}, tile, win, lane);
    else if (ps.kind == SPX_PASS_GATHER_LDS) run_units<1, 2>(a, rb, {ps}, tile, win, lane);
    else run_units<1, 0>(a, rb, {ps}, tile, win, lane);
}

// ---- pass headers as six dwords (the pipelined kernels: spmv_xw_kernels.hip, spmv_sx_kernels.hip) ----
// Read through the constant address space -- the stream is never written while a product runs, and only so
// does the compiler keep fetching them with scalar loads once a kernel contains LDS DMA -- and as whole
// dwords: a byte field read on its own becomes a VECTOR byte load (gfx950 has no scalar one).
typedef const __attribute__((address_space(4))) uint32_t *spx_const_words_t;
struct PassWords {
    uint32_t w[6];
    __device__ __forceinline__ uint64_t mask() const { return (uint64_t) w[0] | ((uint64_t) w[1] << 32); }
    __device__ __forceinline__ uint32_t val_off() const { return w[2]; }
    __device__ __forceinline__ uint32_t rank0() const { return w[3] & 0xffffu; }
    __device__ __forceinline__ uint32_t seg0() const { return w[3] >> 16; }
    __device__ __forceinline__ uint32_t nseg() const { return w[4] & 0xffu; }
    __device__ __forceinline__ uint32_t width() const { return (w[4] >> 8) & 0xffu; }
    __device__ __forceinline__ uint32_t kind() const { return (w[4] >> 16) & 0xffu; }
    __device__ __forceinline__ uint32_t flags() const { return w[4] >> 24; }
    __device__ __forceinline__ SpxPass pass() const
    {
        SpxPass ps;
        ps.mask = mask(); ps.val_off = w[2]; ps.rank0 = (uint16_t) rank0(); ps.seg0 = (uint16_t) seg0();
        ps.nseg = (uint8_t) nseg(); ps.width = (uint8_t) width(); ps.kind = (uint8_t) kind();
        ps.flags = (uint8_t) flags(); ps.elem0 = w[5];
        return ps;
    }
};
static_assert(sizeof(SpxPass) == 24 && offsetof(SpxPass, val_off) == 8 && offsetof(SpxPass, rank0) == 12 &&
              offsetof(SpxPass, seg0) == 14 && offsetof(SpxPass, nseg) == 16 && offsetof(SpxPass, width) == 17 &&
              offsetof(SpxPass, kind) == 18 && offsetof(SpxPass, flags) == 19 && offsetof(SpxPass, elem0) == 20,
              "PassWords mirrors SpxPass");
__device__ __forceinline__ PassWords load_pass(spx_const_words_t passes, int index)
{
    const spx_const_words_t p = passes + 6 * index;
    PassWords h;
#pragma unroll
    for (int k = 0; k < 6; ++k) h.w[k] = p[k];
    return h;
}

// the pass headers of the row-block in LDS (the workgroup copies them there in its prologue: one coalesced
// load instead of a scalar load from memory per pass and wavefront): entry `index` as six dwords, the same
// for every lane, then into SGPRs
__device__ __forceinline__ PassWords lds_pass(const uint32_t *hdr, int index)
{
    const uint2 *p = reinterpret_cast<const uint2 *>(hdr + 6 * index);
    const uint2 a = p[0], b = p[1], c = p[2];
    PassWords h;
    h.w[0] = (uint32_t) __builtin_amdgcn_readfirstlane((int) a.x);
    h.w[1] = (uint32_t) __builtin_amdgcn_readfirstlane((int) a.y);
    h.w[2] = (uint32_t) __builtin_amdgcn_readfirstlane((int) b.x);
    h.w[3] = (uint32_t) __builtin_amdgcn_readfirstlane((int) b.y);
    h.w[4] = (uint32_t) __builtin_amdgcn_readfirstlane((int) c.x);
    h.w[5] = (uint32_t) __builtin_amdgcn_readfirstlane((int) c.y);
    return h;
}

// a pass that is not there (the second half of a round at the end of a wavefront's list): the first
// one's addresses, no lanes
__device__ __forceinline__ PassWords no_pass(const PassWords &like)
{
    PassWords h = like;
    h.w[4] &= ~0xffu;
    return h;
}

#define SPX_KERNEL_PARAMS                                                                        \
    const SpxRowBlock *rbs_, const SpxPass *passes_, uint32_t n_rb_, uint32_t pass_stride_,      \
    XcdSplit xcd_split, const double *values_, const SpxUnitDesc *descs_, \
    const uint8_t *cidx_, const uint16_t *segrows_, const double *x_, double *y_,               \
    double *carry_, const double *dvalues_, double *spill_, const uint32_t *slot_col_,         \
    double alpha_, double beta_, const double *dvalues_priv_, double beta_priv_
#define SPX_KERNEL_ARGS(a)                                                                       \
    KernelArgs a;                                                                                \
    a.rbs = rbs_; a.passes = passes_; a.n_rb = n_rb_; a.pass_stride = pass_stride_;              \
    a.values = values_; a.descs = descs_; a.cidx = cidx_; a.segrows = segrows_; a.x = x_;        \
    a.y = y_; a.carry = carry_; a.dvalues = dvalues_; a.spill = spill_; a.slot_col = slot_col_;  \
    a.alpha = alpha_; a.dvalues_priv = dvalues_priv_; a.beta_priv = beta_priv_;                  \
    a.beta = beta_; a.xw_tab = nullptr

}  // namespace spx
