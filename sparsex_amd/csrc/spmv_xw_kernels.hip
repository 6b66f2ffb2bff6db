// spmv_xw_kernels.hip -- the general-path product with the unit windows of x staged in LDS and the unit
// passes software-pipelined (csx_spmv_xw_kernel; window plan: xwindows.hpp; launched by device_spmv in
// spmv_kernels.hip where the launch tuner found it faster than the plain kernel).
//
// Semantics as the plain kernel's: the reference's SpMV templates (src/templates/csx_spmv_tmpl.c:66-101,
// horiz_tmpl.c:20-37, diag_tmpl.c:20-35, block_row_tmpl.c, block_col_tmpl.c), every stored nonzero a(r,c)
// contributes alpha * a * x[c] to y[r].
#include "spmv_device.hpp"

#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace spx {

// ---- general path, unit windows of x in LDS, unit passes software-pipelined ------------------------
//
// What the plain kernel above leaves on the table on a large streaming matrix (profiles/r04/spread.md:
// the bare read probe holds 5.9 TB/s on every node, this product swings by 16 % with the node): a
// wavefront runs header -> {values, x} -> FMAs -> LDS adds strictly in turn, so between two passes it has
// nothing in flight, and every pass pushes as many bytes of x through the vector L1 as it reads values.
// Here (csx_spmv_xw_kernel):
//  * the workgroup stages the row-block's unit windows (xwindows.hpp) with global_load_lds_dwordx4 --
//    asynchronous, coalesced, no registers -- and every unit pass that carries SPX_PASSF_XLDS reads its x
//    with ds_read from there: no dependent global load, no x traffic in the L1;
//  * pairs of unit passes of equal width run as a two-stage pipeline: the values (and descriptors) of
//    the NEXT pair are requested before the FMAs and LDS adds of the current one, pass headers are
//    fetched two rounds ahead -- a wavefront always has one or two pairs of passes in flight.
// Everything else (gather passes, unit passes of row-blocks whose columns did not fit the window budget,
// single passes at the end of a wavefront's list) runs through the code of the plain kernel.
// Pass headers are read through the constant address space -- the stream is never written while a product
// runs, and only so does the compiler keep fetching them with scalar loads (s_load into SGPRs, out of the
// way of the vector memory counter) once the kernel contains LDS DMA, which it must assume writes memory --
// and as six whole dwords: a byte field read on its own becomes a VECTOR byte load (gfx950 has no scalar
// one), and the wait for it is a wait for every value load in flight (PassWords, spmv_device.hpp).

// One stage of the pipeline: B unit passes of ANY width 1..4, three loads each -- the lane's descriptor
// and two 16-byte loads at 8-byte granularity that between them hold its values whatever the width
// (W = 1: {v0, -}; 2: {v0, v1}; 3: {v0, v1}, {v2, -}; 4: {v0, v1}, {v2, v3}; "-" is whatever follows in
// the value array, never used).  The same number of loads for every pass is what lets passes of
// different widths follow each other in one pipeline, and lets the compiler count its loads.
template <int B>
struct XwStage {
    uint2 q[B];                // {offset of segment 0 in the unit windows, descriptor bits}
    uint32_t segl[B];          // segments of the row-block in front of the lane's | lane active << 16
    uint32_t width[B];         // (wave-uniform)
    uint32_t row0[B];          // (wave-uniform) first row of the pass' part of the row-block (SpxPass::elem0)
    spx_d2u_t va[B], vb[B];
};

// The loads of a stage.  Everything the second half needs of the headers goes into the stage with them:
// the headers are dead once their loads are out.  (The descriptor comes from the descriptor array also
// where the header holds a copy: x comes from LDS once the values are there, so the copy would save no
// round trip here.)
template <int B>
__device__ __forceinline__ void xw_issue(const KernelArgs &a, const SpxRowBlock &rb, const PassWords (&ps)[B],
                                         XwStage<B> &S, int lane)
{
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const uint32_t nseg = ps[b].nseg(), W = ps[b].width();
        const bool active = (uint32_t) lane < nseg;
        const uint32_t l = active ? (uint32_t) lane : 0u;
        S.segl[b] = ps[b].seg0() + l + (active ? 0x10000u : 0u);
        S.width[b] = W;
        S.row0[b] = ps[b].w[5];
        const uint64_t mk = (ps[b].flags() & SPX_PASSF_INLINE) ? 0ull : ps[b].mask();
        const uint32_t rank = ps[b].rank0() + (active ? starts_upto(mk, lane) : 0u);
        S.q[b] = ld_stream(reinterpret_cast<const uint2 *>(a.descs + rb.desc_off + rank));
        const double *vals = a.values + rb.val_off + ps[b].val_off();
        const uint32_t off_a = W == 1u ? l : 2u * l;
        const uint32_t off_b = W == 3u ? 2u * nseg + l : (W == 4u ? 2u * nseg + 2u * l : off_a);
        S.va[b] = *reinterpret_cast<const spx_d2u_t *>(vals + off_a);
        S.vb[b] = *reinterpret_cast<const spx_d2u_t *>(vals + off_b);
    }
}

// ... and what follows once they have arrived: rows and window offsets, x from LDS, W FMAs, one LDS add
template <int W>
__device__ __forceinline__ void xw_finish_pass(uint2 q, uint32_t segl, uint32_t row0, spx_d2u_t va, spx_d2u_t vb,
                                               double *tile, const double *xw)
{
    const uint32_t c0 = q.x, bits = q.y;
    const int s = (int) ((segl - ((bits >> 9) & 8191u)) & 0xffffu);
    const uint32_t kind = (bits >> 22) & 7u;
    const int step = (int) (bits >> 25);
    const int drow = kind == SPX_KIND_BLOCK ? 1 : (kind >= SPX_KIND_VERT ? step : 0);
    const int dcol = (kind == SPX_KIND_HORIZ || kind == SPX_KIND_DIAG)
                         ? step : (kind == SPX_KIND_ADIAG ? -step : 0);
    const int row = (int) (row0 + (bits & 511u)) + s * drow;
    const double *xp = xw + (int) (c0 + (uint32_t) (s * dcol));      // (c0: an offset into the unit windows)
    double t = va.x * xp[0];
    if (W >= 2) t = fma(va.y, xp[1], t);
    if (W >= 3) t = fma(vb.x, xp[2], t);
    if (W >= 4) t = fma(vb.y, xp[3], t);
    if (segl >> 16) atomicAdd(&tile[row], t);
}

template <int B>
__device__ __forceinline__ void xw_finish(const XwStage<B> &S, double *tile, const double *xw)
{
#pragma unroll
    for (int b = 0; b < B; ++b) {
        switch (S.width[b]) {          // wave-uniform
        case 1: xw_finish_pass<1>(S.q[b], S.segl[b], S.row0[b], S.va[b], S.vb[b], tile, xw); break;
        case 2: xw_finish_pass<2>(S.q[b], S.segl[b], S.row0[b], S.va[b], S.vb[b], tile, xw); break;
        case 3: xw_finish_pass<3>(S.q[b], S.segl[b], S.row0[b], S.va[b], S.vb[b], tile, xw); break;
        default: xw_finish_pass<4>(S.q[b], S.segl[b], S.row0[b], S.va[b], S.vb[b], tile, xw); break;
        }
    }
}

// a unit pass of width 5..8 that reads LDS, on its own (not pipelined)
template <int W>
__device__ __forceinline__ void xw_wide(const KernelArgs &a, const SpxRowBlock &rb, const PassWords &ps,
                                        double *tile, const double *xw, int lane)
{
    const uint32_t nseg = ps.nseg();
    const bool active = (uint32_t) lane < nseg;
    const uint32_t l = active ? (uint32_t) lane : 0u;
    const uint64_t mk = (ps.flags() & SPX_PASSF_INLINE) ? 0ull : ps.mask();
    const uint32_t rank = ps.rank0() + (active ? starts_upto(mk, lane) : 0u);
    const uint2 q = ld_stream(reinterpret_cast<const uint2 *>(a.descs + rb.desc_off + rank));
    const double *vals = a.values + rb.val_off + ps.val_off();
    double2 v2[W / 2];
    double v1 = 0.0;
#pragma unroll
    for (int p = 0; p < W / 2; ++p)
        v2[p] = ld_stream(reinterpret_cast<const double2 *>(vals + (uint32_t) p * 2u * nseg + l * 2u));
    if (W & 1) v1 = ld_stream(vals + (uint32_t) (W / 2) * 2u * nseg + l);
    const uint32_t bits = q.y;
    const int s = (int) ((ps.seg0() + l - ((bits >> 9) & 8191u)) & 0xffffu);
    const uint32_t kind = (bits >> 22) & 7u;
    const int step = (int) (bits >> 25);
    const int drow = kind == SPX_KIND_BLOCK ? 1 : (kind >= SPX_KIND_VERT ? step : 0);
    const int dcol = (kind == SPX_KIND_HORIZ || kind == SPX_KIND_DIAG) ? step : (kind == SPX_KIND_ADIAG ? -step : 0);
    const int row = (int) (ps.w[5] + (bits & 511u)) + s * drow;
    const double *xp = xw + (int) (q.x + (uint32_t) (s * dcol));
    double t = 0.0;
#pragma unroll
    for (int p = 0; p < W / 2; ++p) {
        t = fma(v2[p].x, xp[2 * p], t);
        t = fma(v2[p].y, xp[2 * p + 1], t);
    }
    if (W & 1) t = fma(v1, xp[W - 1], t);
    if (active) atomicAdd(&tile[row], t);
}

// one pass of any kind on its own
__device__ __forceinline__ void xw_one(const KernelArgs &a, const SpxRowBlock &rb, const PassWords &ps,
                                       double *tile, const double *win, const double *xw, int lane)
{
    if (ps.kind() == SPX_PASS_UNIT && (ps.flags() & SPX_PASSF_XLDS)) {
        if (ps.width() <= 4u) {
            XwStage<1> S;
            xw_issue<1>(a, rb, {ps}, S, lane);
            xw_finish<1>(S, tile, xw);
            return;
        }
        switch (ps.width()) {            // wave-uniform
        case 5: xw_wide<5>(a, rb, ps, tile, xw, lane); break;
        case 6: xw_wide<6>(a, rb, ps, tile, xw, lane); break;
        case 7: xw_wide<7>(a, rb, ps, tile, xw, lane); break;
        default: xw_wide<8>(a, rb, ps, tile, xw, lane); break;
        }
    } else {
        run_pass(a, rb, ps.pass(), tile, win, lane);
    }
}

// A wavefront takes the passes t, t + WAVES, t + 2 WAVES, ... of its row-block.  Those of them that lie in
// the row-block's range [lo, hi) of narrow unit passes that read LDS (xwindows.hpp) run through the
// pipeline, two to a round (an odd one out at the end shares its round with an empty pass): how many of
// them follow from pass t on
template <int WAVES>
__device__ __forceinline__ int xw_in_range(int t, int lo, int hi)
{
    return (t >= lo && t < hi) ? (hi - 1 - t) / WAVES + 1 : 0;
}

// Runs the `n_in` >= 1 passes t, t + WAVES, ... in rounds of two as a two-stage pipeline: the loads of the
// next round go out before the FMAs and LDS adds of the current one.  `A` holds the loads of the first
// round, already issued.  On return t is the wavefront's next pass.
// (A counted loop without a branch around any load: the compiler counts outstanding loads per path, and
// with the loads of the next round under a condition it waited for every load before every use, which
// undid the pipeline.  Deeper pipelines were measured -- three and four rounds in flight, with empty rounds
// at the ends so that no load sits under a branch: slower on the five rounds a wavefront has per row-block,
// and what they gain in flight they lose in wavefronts per SIMD: profiles/r05/ablation.md.)
template <int WAVES>
__device__ __forceinline__ void xw_run(const KernelArgs &a, const SpxRowBlock &rb, const uint32_t *hdr, int hi,
                                       int n_in, int &t, XwStage<2> &A, double *tile, const double *xw, int lane)
{
    XwStage<2> B;
    PassWords c0, c1;
    const int n_rounds = (n_in + 1) / 2, t_end = t + n_in * WAVES;
    // (headers of the next round: read from LDS while the current round's second half runs)
#define SPX_XW_HEADERS()                                                                          \
    do {                                                                                          \
        t += 2 * WAVES;                                                                           \
        c0 = lds_pass(hdr, t);                                                                    \
        c1 = lds_pass(hdr, t + WAVES);                                                            \
        if (t + WAVES >= hi) c1 = no_pass(c0);                                                    \
    } while (0)
    int r = 1;
    for (; r + 1 < n_rounds; r += 2) {
        SPX_XW_HEADERS();
        xw_issue<2>(a, rb, {c0, c1}, B, lane);
        xw_finish<2>(A, tile, xw);
        SPX_XW_HEADERS();
        xw_issue<2>(a, rb, {c0, c1}, A, lane);
        xw_finish<2>(B, tile, xw);
    }
    if (r < n_rounds) {
        SPX_XW_HEADERS();
        xw_issue<2>(a, rb, {c0, c1}, B, lane);
        xw_finish<2>(A, tile, xw);
        xw_finish<2>(B, tile, xw);
    } else {
        xw_finish<2>(A, tile, xw);
    }
    t = t_end;
#undef SPX_XW_HEADERS
}

template <int WAVES>
__device__ __forceinline__ void spmv_body_xw(const KernelArgs &a, const XcdSplit &xs, double *lds)
{
    constexpr int BLOCK_THREADS = 64 * WAVES;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t rb_idx = xs.first[xcd] + (blockIdx.x >> 3);
    if (rb_idx >= xs.first[xcd + 1u]) return;

    // first round trip: the row-block header, the table of its unit windows (one entry per lane; handed to
    // the whole wavefront with v_readlane), the wavefront's first two pass headers (scalar), and -- on their
    // way to LDS -- all pass headers of the row-block (pass_stride of them whatever the row-block uses:
    // no need to wait for its header to know how many)
    const SpxPass *pass0 = a.passes + (size_t) rb_idx * a.pass_stride;
    spx_const_words_t passes = (spx_const_words_t) (uintptr_t) pass0;
    const SpxRowBlock rb = a.rbs[rb_idx];
    const uint2 xw_entry = *reinterpret_cast<const uint2 *>(a.xw_tab + (size_t) rb_idx * XW_TAB + (lane & (XW_TAB - 1)));
    PassWords c0 = load_pass(passes, wave), c1 = load_pass(passes, wave + WAVES);       // (the table is padded)
    const int n_rows = rb.n_rows;
    double *tile = lds;
    for (int i = threadIdx.x; i < n_rows; i += BLOCK_THREADS) lds[i] = 0.0;
    double *win = lds + n_rows;
    {
        const int xwl = rb.xwin_len;
        const double *xsrc = a.x + rb.xwin_base;
        for (int i = threadIdx.x; i < xwl; i += BLOCK_THREADS) win[i] = xsrc[i];
    }
    // the unit windows: every wavefront moves pieces of 128 doubles, 16 bytes per lane, straight into LDS
    // (global_load_lds_dwordx4: asynchronous, no registers; lane i's 16 bytes land at the wavefront's
    // LDS address + 16 i)
    double *xw = lds + ((n_rows + (int) rb.xwin_len + 1) & ~1);
    const uint32_t range = (uint32_t) __builtin_amdgcn_readlane((int) xw_entry.x, 0);
    const uint32_t xw_total = (uint32_t) __builtin_amdgcn_readlane((int) xw_entry.y, 0);
    const int lo = (int) (range & 0xffffu), hi = (int) (range >> 16);
    uint32_t *hdr = reinterpret_cast<uint32_t *>(xw + xw_total);
    uint32_t odd_base = 0, odd_at = 0xffffffffu;
#pragma unroll
    for (uint32_t k = 0; k < XW_MAX; ++k) {
        const uint32_t base = (uint32_t) __builtin_amdgcn_readlane((int) xw_entry.x, (int) (XW_RANGES + k));
        const uint32_t off_len = (uint32_t) __builtin_amdgcn_readlane((int) xw_entry.y, (int) (XW_RANGES + k));
        const uint32_t len = off_len >> 16, off = off_len & 0xffffu;
        if (len == 0) break;
        const double *src = a.x + base;
        for (uint32_t c = (uint32_t) wave * 128u; c + 1u < len; c += (uint32_t) WAVES * 128u) {
            const uint32_t i = c + 2u * (uint32_t) lane;
            if (i + 1u < len)
                __builtin_amdgcn_global_load_lds(src + i, (__attribute__((address_space(3))) void *) (xw + off + c), 16, 0, 0);
        }
        if (len & 1u) {            // (a window that ends with a vector of odd length: its last double)
            odd_base = base + len - 1u;
            odd_at = off + len - 1u;
        }
    }
    if (odd_at != 0xffffffffu && threadIdx.x == 0) xw[odd_at] = a.x[odd_base];
    // ... and the pass headers behind them, the same way: pass_stride + 4 WAVES of them (the table is padded),
    // a kilobyte per wavefront and step
    {
        const uint32_t n_words = 6u * (a.pass_stride + 4u * (uint32_t) WAVES);
        const uint32_t *src = reinterpret_cast<const uint32_t *>(pass0);
        for (uint32_t c = (uint32_t) wave * 256u; c < n_words; c += (uint32_t) WAVES * 256u) {
            const uint32_t i = c + 4u * (uint32_t) lane;
            if (i < n_words)
                __builtin_amdgcn_global_load_lds(src + i, (__attribute__((address_space(3))) void *) (hdr + c), 16, 0, 0);
        }
    }
    // the loads of the wavefront's first round go out in front of the barrier, next to the windows
    const int n_pass = rb.n_pass;
    int t = wave;
    const int n_first = xw_in_range<WAVES>(t, lo, hi);
    XwStage<2> A;
    if (n_first > 0) {
        if (t + WAVES >= hi) c1 = no_pass(c0);
        xw_issue<2>(a, rb, {c0, c1}, A, lane);
    }
    __syncthreads();

    if (n_first > 0) xw_run<WAVES>(a, rb, hdr, hi, n_first, t, A, tile, xw, lane);
    while (t < n_pass) {
        const int n_in = xw_in_range<WAVES>(t, lo, hi);
        c0 = lds_pass(hdr, t);
        c1 = lds_pass(hdr, t + WAVES);
        if (n_in > 0) {
            if (t + WAVES >= hi) c1 = no_pass(c0);
            xw_issue<2>(a, rb, {c0, c1}, A, lane);
            xw_run<WAVES>(a, rb, hdr, hi, n_in, t, A, tile, xw, lane);
            continue;
        }
        xw_one(a, rb, c0, tile, win, xw, lane);
        if (t + WAVES < n_pass) xw_one(a, rb, c1, tile, win, xw, lane);
        t += 2 * WAVES;
    }
    __syncthreads();

    if (rb.flags & SPX_RB_SHARED) {
        if (threadIdx.x == 0) a.carry[rb.carry_slot] = tile[0];
    } else {
        for (int i = threadIdx.x; i < n_rows; i += BLOCK_THREADS) {
            const size_t g = (size_t) rb.row0 + i;
            double tt = a.alpha * tile[i];
            if (a.beta != 0.0) tt += a.beta * a.y[g];
            a.y[g] = tt;
        }
    }
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES)
void csx_spmv_xw_kernel(SPX_KERNEL_PARAMS, const XwEntry *xw_tab_)
{
    SPX_KERNEL_ARGS(a);
    a.xw_tab = xw_tab_;
    extern __shared__ double lds_dyn[];      // y tile, the leftovers' x window, the unit windows, the pass headers
    spmv_body_xw<WAVES>(a, xcd_split, lds_dyn);
}

void launch_spmv_xw(int waves, unsigned blocks, size_t lds_bytes, void *stream_, const KernelArgs &a, const XcdSplit &xs)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
#define SPX_LAUNCH_XW(W)                                                                           \
    hipLaunchKernelGGL(csx_spmv_xw_kernel<W>, dim3(blocks), dim3(64 * W), lds_bytes, stream, a.rbs,  \
                       a.passes, a.n_rb, a.pass_stride, xs, a.values, a.descs, a.cidx, a.segrows,   \
                       a.x, a.y, a.carry, a.dvalues, a.spill, a.slot_col, a.alpha, a.beta,           \
                       a.dvalues_priv, a.beta_priv, a.xw_tab)
    if (waves == 2) SPX_LAUNCH_XW(2);
    else if (waves == 8) SPX_LAUNCH_XW(8);
    else SPX_LAUNCH_XW(4);
#undef SPX_LAUNCH_XW
}

// row-blocks whose windows need more than the default 64 KB of dynamic LDS
void spmv_xw_allow_lds(size_t bytes)
{
    const int b = (int) bytes;
    (void) hipFuncSetAttribute(reinterpret_cast<const void *>(&csx_spmv_xw_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, b);
    (void) hipFuncSetAttribute(reinterpret_cast<const void *>(&csx_spmv_xw_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, b);
    (void) hipFuncSetAttribute(reinterpret_cast<const void *>(&csx_spmv_xw_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, b);
}

}  // namespace spx
