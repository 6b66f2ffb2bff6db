// spmv_sx_kernels.hip -- the symmetric read-once product with its passes software-pipelined
// (csx_spmv_sx_kernel; device-side headers: sxplan.hpp; launched by device_spmv in spmv_kernels.hip for
// streams of read-once row segments without tiles, where the launch tuner found it faster).
//
// Semantics as csx_spmv_symseg_notile_kernel's: the reference's symmetric SpMV template
// (src/templates/csx_sym_spmv_tmpl.c:60-106; unit bodies horiz_sym_tmpl.c, diag_sym_tmpl.c, ...): a stored
// a(r,c) of the strictly lower triangle contributes alpha * a * x[c] to y[r] and alpha * a * x[r] to y[c]
// (`cur[c] += x[r] * v * alpha`, :92-95); dispatch src/internals/CsxKernels.cpp:105-129.
//
// What the plain read-once kernel leaves on the table (profiles/r05/ablation.md section 2: with x, slot adds
// and hand-over all removed it still streams at 4.0 TB/s): a wavefront runs pass header -> descriptors ->
// {values, x} -> FMAs -> LDS adds strictly in turn, three dependent memory round trips per round of two
// passes and nothing in flight in between.  Here
//  * the pass headers of the row-block sit in LDS (copied there with global_load_lds_dwordx4 in the prologue);
//  * an SX pass (all lanes one unit, geometry in the header: sxplan.hpp) asks for its values, x[row] and
//    x[columns] in ONE round trip -- no descriptor load at all;
//  * rounds of B SX passes (B = SX_PASSES_PER_ROUND = 1: measured, see below) run as a two-stage pipeline: the
//    loads of the NEXT round go out before the FMAs and LDS adds of the current one.
// x comes through the vector memory path, not from LDS (the unit-window kernel's way): a row-block's slots
// and y tile already take 44 KB, a window of x of the same shape would halve the workgroups per CU.  Loads
// return in order, so x must travel WITH the values of its own round (a dependent x load behind the next
// round's values would wait for them) -- which is exactly what the header-resident geometry allows.
#include "spmv_sym_device.hpp"
#include "sxplan.hpp"

#include <cstddef>

namespace spx {

// One stage of the pipeline: B SX passes of any width 1..4, five loads each -- two 16-byte loads that between
// them hold the lane's values whatever the width (as in the unit-window kernel: W = 1: {v0, -}; 2: {v0, v1};
// 3: {v0, v1}, {v2, -}; 4: {v0, v1}, {v2, v3}), x[row], and two 16-byte loads of x[col ...] (the second one
// repeats the first where W <= 2: a strictly lower segment never reaches x[row], so col + 3 <= row stays
// inside x for W >= 3, col + 1 <= row for W = 1).  The same number of loads for every pass lets passes of
// different widths follow each other in one pipeline, and lets the compiler count its loads.
template <int B>
struct SxStage {
    uint32_t col[B];           // (wave-uniform) first column of lane 0's segment
    uint32_t geo[B];           // (wave-uniform) row of lane 0 | drow << 11 | (dcol + 128) << 18
    uint32_t slot[B];          // (wave-uniform) slot of lane 0's first column, or SPX_NO_SLOT
    uint32_t nw[B];            // (wave-uniform) nseg | width << 8
    spx_d2u_t va[B], vb[B], xa[B], xb[B];
    double xr[B];
};

template <int B>
__device__ __forceinline__ void sx_issue(const KernelArgs &a, const SpxRowBlock &rb, const PassWords (&ps)[B],
                                         SxStage<B> &S, int lane)
{
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const uint32_t nseg = ps[b].nseg(), W = ps[b].width();
        const uint32_t l = (uint32_t) lane < nseg ? (uint32_t) lane : 0u;      // idle lanes shadow lane 0
        const uint32_t geo = ps[b].w[1];
        S.col[b] = ps[b].w[0];
        S.geo[b] = geo;
        S.slot[b] = ps[b].w[3];
        S.nw[b] = ps[b].w[4] & 0xffffu;
        const double *vals = a.values + rb.val_off + ps[b].val_off();
        const uint32_t off_a = W == 1u ? l : 2u * l;
        const uint32_t off_b = W == 3u ? 2u * nseg + l : (W == 4u ? 2u * nseg + 2u * l : off_a);
        S.va[b] = *reinterpret_cast<const spx_d2u_t *>(vals + off_a);
        S.vb[b] = *reinterpret_cast<const spx_d2u_t *>(vals + off_b);
        const int drow = (int) ((geo >> 11) & 127u), dcol = (int) ((geo >> 18) & 255u) - 128;
        const uint32_t row = (geo & 2047u) + l * (uint32_t) drow;
        const double *xp = a.x + (S.col[b] + (uint32_t) ((int) l * dcol));
        if (abl::sym_no_x) {
            S.xr[b] = a.x[lane];
            S.xa[b] = *reinterpret_cast<const spx_d2u_t *>(a.x + lane);
            S.xb[b] = S.xa[b];
        } else {
            S.xr[b] = a.x[rb.row0 + row];
            S.xa[b] = *reinterpret_cast<const spx_d2u_t *>(xp);
            S.xb[b] = *reinterpret_cast<const spx_d2u_t *>(xp + (W >= 3u ? 2 : 0));
        }
    }
}

// ... and what follows once they have arrived: W FMAs and one LDS add for the row, W LDS adds of the
// transposed products for the columns (or, without slots, W global atomics)
template <int W>
__device__ __forceinline__ void sx_finish_pass(const KernelArgs &a, uint32_t col0, uint32_t geo, uint32_t slot0,
                                               uint32_t nw, spx_d2u_t va, spx_d2u_t vb, spx_d2u_t xa, spx_d2u_t xb,
                                               double xr, double *slots, double *tile, int lane)
{
    double t = va.x * xa.x;
    if (W >= 2) t = fma(va.y, xa.y, t);
    if (W >= 3) t = fma(vb.x, xb.x, t);
    if (W >= 4) t = fma(vb.y, xb.y, t);
    if ((uint32_t) lane >= (nw & 0xffu)) return;
    const int drow = (int) ((geo >> 11) & 127u), dcol = (int) ((geo >> 18) & 255u) - 128;
    const uint32_t row = (geo & 2047u) + (uint32_t) lane * (uint32_t) drow;
    const int sdc = lane * dcol;
    atomicAdd(&tile[row], t);
    if (slot0 != SPX_NO_SLOT) {            // (wave-uniform)
        double *sl = slots + (int) slot0 + sdc;
        if (abl::sym_no_slot_add) {
            double u = va.x * xr;
            if (W >= 2) u += va.y * xr;
            if (W >= 3) u += vb.x * xr;
            if (W >= 4) u += vb.y * xr;
            if (u == 1.2345e-300) sl[0] = u;
        } else {
            atomicAdd(&sl[0], va.x * xr);
            if (W >= 2) atomicAdd(&sl[1], va.y * xr);
            if (W >= 3) atomicAdd(&sl[2], vb.x * xr);
            if (W >= 4) atomicAdd(&sl[3], vb.y * xr);
        }
    } else {
        double *yp = a.y + (col0 + (uint32_t) sdc);
        atomicAdd(&yp[0], a.alpha * (va.x * xr));
        if (W >= 2) atomicAdd(&yp[1], a.alpha * (va.y * xr));
        if (W >= 3) atomicAdd(&yp[2], a.alpha * (vb.x * xr));
        if (W >= 4) atomicAdd(&yp[3], a.alpha * (vb.y * xr));
    }
}

template <int B>
__device__ __forceinline__ void sx_finish(const KernelArgs &a, const SxStage<B> &S, double *slots, double *tile, int lane)
{
#pragma unroll
    for (int b = 0; b < B; ++b) {
        switch ((S.nw[b] >> 8) & 0xffu) {          // wave-uniform
        case 1: sx_finish_pass<1>(a, S.col[b], S.geo[b], S.slot[b], S.nw[b], S.va[b], S.vb[b], S.xa[b], S.xb[b], S.xr[b], slots, tile, lane); break;
        case 2: sx_finish_pass<2>(a, S.col[b], S.geo[b], S.slot[b], S.nw[b], S.va[b], S.vb[b], S.xa[b], S.xb[b], S.xr[b], slots, tile, lane); break;
        case 3: sx_finish_pass<3>(a, S.col[b], S.geo[b], S.slot[b], S.nw[b], S.va[b], S.vb[b], S.xa[b], S.xb[b], S.xr[b], slots, tile, lane); break;
        default: sx_finish_pass<4>(a, S.col[b], S.geo[b], S.slot[b], S.nw[b], S.va[b], S.vb[b], S.xa[b], S.xb[b], S.xr[b], slots, tile, lane); break;
        }
    }
}

// Runs the wavefront's `n_in` >= 1 SX passes t, t + WAVES, ... in rounds of B as a two-stage pipeline (the
// structure of xw_run, spmv_xw_kernels.hip: a counted loop without a branch around any load, so that the
// compiler's load counting keeps one round in flight behind the one being finished).  `A` holds the loads of
// the first round, already issued; a round that the passes do not fill is topped up with empty ones.
template <int WAVES, int B>
__device__ __forceinline__ void sx_headers(const uint32_t *hdr, int hi, int t, PassWords (&c)[B])
{
    c[0] = lds_pass(hdr, t);
#pragma unroll
    for (int b = 1; b < B; ++b) {
        c[b] = lds_pass(hdr, t + b * WAVES);
        if (t + b * WAVES >= hi) c[b] = no_pass(c[0]);
    }
}

template <int WAVES, int B>
__device__ __forceinline__ void sx_run(const KernelArgs &a, const SpxRowBlock &rb, const uint32_t *hdr, int hi,
                                       int n_in, int &t, SxStage<B> &A, double *slots, double *tile, int lane)
{
    SxStage<B> N;
    PassWords c[B];
    const int n_rounds = (n_in + B - 1) / B, t_end = t + n_in * WAVES;
    int r = 1;
    for (; r + 1 < n_rounds; r += 2) {
        t += B * WAVES;
        sx_headers<WAVES, B>(hdr, hi, t, c);
        sx_issue<B>(a, rb, c, N, lane);
        sx_finish<B>(a, A, slots, tile, lane);
        t += B * WAVES;
        sx_headers<WAVES, B>(hdr, hi, t, c);
        sx_issue<B>(a, rb, c, A, lane);
        sx_finish<B>(a, N, slots, tile, lane);
    }
    if (r < n_rounds) {
        t += B * WAVES;
        sx_headers<WAVES, B>(hdr, hi, t, c);
        sx_issue<B>(a, rb, c, N, lane);
        sx_finish<B>(a, A, slots, tile, lane);
        sx_finish<B>(a, N, slots, tile, lane);
    } else {
        sx_finish<B>(a, A, slots, tile, lane);
    }
    t = t_end;
}

// a pass that is not an SX pass, on its own: the code of the plain kernels
__device__ __forceinline__ void sx_other(const KernelArgs &a, const SpxRowBlock &rb, const SpxPass &ps, double *slots,
                                         double *tile, const double *win, int lane)
{
    if (ps.kind == SPX_PASS_SYMSEG) {
        if (!abl::sym_no_mixed) run_symseg(a, rb, ps, slots, tile, lane);
    } else {
        run_pass(a, rb, ps, tile, win, lane);
    }
}

template <int WAVES, int B>
__device__ __forceinline__ void spmv_body_sx(const KernelArgs &a, const XcdSplit &xs, const uint32_t *sx_tab, double *lds)
{
    constexpr int BLOCK_THREADS = 64 * WAVES;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t rb_idx = xs.first[xcd] + (blockIdx.x >> 3);
    if (rb_idx >= xs.first[xcd + 1u]) return;

    // first round trip: the row-block header, the number of its SX passes, the headers of the wavefront's
    // first round (scalar), and -- on their way to LDS -- all pass headers of the row-block (pass_stride of them
    // whatever the row-block uses: no need to wait for its header to know how many)
    const SpxPass *pass0 = a.passes + (size_t) rb_idx * a.pass_stride;
    spx_const_words_t passes = (spx_const_words_t) (uintptr_t) pass0;
    const SpxRowBlock rb = a.rbs[rb_idx];
    const int hi = (int) sx_tab[rb_idx];                 // the passes [0, hi) are SX passes
    PassWords c[B];
#pragma unroll
    for (int b = 0; b < B; ++b) c[b] = load_pass(passes, wave + b * WAVES);             // (the table is padded)
    const int n_rows = rb.n_rows, n_slots = (int) rb.n_slots;
    const int core = n_slots + n_rows;
    double *slots = lds, *tile = lds + n_slots;
    for (int i = threadIdx.x; i < core; i += BLOCK_THREADS) lds[i] = 0.0;
    double *win = lds + core;
    {
        const int xwl = rb.xwin_len;
        const double *xsrc = a.x + rb.xwin_base;
        for (int i = threadIdx.x; i < xwl; i += BLOCK_THREADS) win[i] = xsrc[i];
    }
    // the first columns of the slot groups, for the hand-over at the end
    uint32_t *gcol_lds = reinterpret_cast<uint32_t *>(win + rb.xwin_len);
    {
        const uint32_t *gcol = a.slot_col + (rb.spill_off >> 3);
        for (int i = threadIdx.x; i < (n_slots >> 3); i += BLOCK_THREADS) gcol_lds[i] = gcol[i];
    }
    // the pass headers behind them (16-byte aligned), straight into LDS: pass_stride + 4 WAVES of them (the
    // table is padded), a kilobyte per wavefront and step
    uint32_t *hdr;
    {
        uint32_t off = (uint32_t) (core + (int) rb.xwin_len) * 8u + (uint32_t) (n_slots >> 3) * 4u;
        off = (off + 15u) & ~15u;
        hdr = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(lds) + off);
    }
    {
        const uint32_t n_words = 6u * (a.pass_stride + 4u * (uint32_t) WAVES);
        const uint32_t *src = reinterpret_cast<const uint32_t *>(pass0);
        for (uint32_t c = (uint32_t) wave * 256u; c < n_words; c += (uint32_t) WAVES * 256u) {
            const uint32_t i = c + 4u * (uint32_t) lane;
            if (i < n_words)
                __builtin_amdgcn_global_load_lds(src + i, (__attribute__((address_space(3))) void *) (hdr + c), 16, 0, 0);
        }
    }
    // the loads of the wavefront's first round go out in front of the barrier
    const int n_pass = rb.n_pass;
    int t = wave;
    const int n_first = t < hi ? (hi - 1 - t) / WAVES + 1 : 0;
    SxStage<B> A;
    if (n_first > 0) {
#pragma unroll
        for (int b = 1; b < B; ++b)
            if (t + b * WAVES >= hi) c[b] = no_pass(c[0]);
        sx_issue<B>(a, rb, c, A, lane);
    }
    __syncthreads();

    if (n_first > 0) sx_run<WAVES, B>(a, rb, hdr, hi, n_first, t, A, slots, tile, lane);
    // what is left: read-once passes of several units, unit passes of the mirrored part, leftovers
    while (t < n_pass) {
        const SpxPass p0 = lds_pass(hdr, t).pass();
        const bool two = t + WAVES < n_pass;
        if (two) {
            const SpxPass p1 = lds_pass(hdr, t + WAVES).pass();
            if (!(p0.kind == SPX_PASS_SYMSEG && p1.kind == SPX_PASS_SYMSEG &&
                  (abl::sym_no_mixed || run_symseg2(a, rb, p0, p1, slots, tile, lane)))) {
                sx_other(a, rb, p0, slots, tile, win, lane);
                sx_other(a, rb, p1, slots, tile, win, lane);
            }
        } else {
            sx_other(a, rb, p0, slots, tile, win, lane);
        }
        t += 2 * WAVES;
    }
    __syncthreads();

    // ---------------- hand-over (as csx_spmv_symseg_notile_kernel) ---------------------------------------
    if (rb.flags & SPX_RB_SHARED) {
        if (threadIdx.x == 0) a.carry[rb.carry_slot] = tile[0];
    } else if (abl::sym_no_own) {
        // (experiment build: the own rows stay where they are)
    } else if ((rb.flags & SPX_RB_PRIVATE) && a.dvalues_priv) {
        // nobody else adds to these rows (mark_private_rowblocks): stored, with the diagonal term and
        // beta * y; the init pass leaves them out
        for (int i = threadIdx.x; i < n_rows; i += BLOCK_THREADS) {
            const size_t g = (size_t) rb.row0 + i;
            double tt = a.alpha * (tile[i] + a.dvalues_priv[g] * a.x[g]);
            if (a.beta_priv != 0.0) tt += a.beta_priv * a.y[g];
            a.y[g] = tt;
        }
    } else {
        for (int i = threadIdx.x; i < n_rows; i += BLOCK_THREADS)
            atomicAdd(&a.y[(size_t) rb.row0 + i], a.alpha * tile[i]);
    }
    if (!abl::sym_no_handover)
        for (int i = threadIdx.x; i < n_slots; i += BLOCK_THREADS)
            atomicAdd(&a.y[(size_t) gcol_lds[i >> 3] + (i & 7)], a.alpha * lds[i]);
}

template <int WAVES, int B>
__global__ __launch_bounds__(64 * WAVES)
void csx_spmv_sx_kernel(SPX_KERNEL_PARAMS, const uint32_t *sx_tab_)
{
    SPX_KERNEL_ARGS(a);
    extern __shared__ double lds_dyn[];      // slots, y tile, the leftovers' x window, slot-group columns, pass headers
    spmv_body_sx<WAVES, B>(a, xcd_split, sx_tab_, lds_dyn);
}

// (B: passes per round of the pipeline.  Two per round -- the unit-window kernel's choice -- take 90 VGPRs, five
// wavefronts per SIMD, and measured 3-4 % slower than one per round at 64 VGPRs: 947 / 939 against 913 / 903 us on
// the bench matrix, profiles/r06/sx_order_and_width_raw.md; three workgroups of eight wavefronts per CU with one
// round in flight behind the one being finished already cover the latency.)
constexpr int SX_PASSES_PER_ROUND = 1;

void launch_spmv_sx(int waves, unsigned blocks, size_t lds_bytes, void *stream_, const KernelArgs &a, const XcdSplit &xs,
                    const uint32_t *sx_tab)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
#define SPX_LAUNCH_SX(W)                                                                              \
    hipLaunchKernelGGL((csx_spmv_sx_kernel<W, SX_PASSES_PER_ROUND>), dim3(blocks), dim3(64 * W), lds_bytes, stream, a.rbs,  \
                       a.passes, a.n_rb, a.pass_stride, xs, a.values, a.descs, a.cidx, a.segrows,   \
                       a.x, a.y, a.carry, a.dvalues, a.spill, a.slot_col, a.alpha, a.beta,           \
                       a.dvalues_priv, a.beta_priv, sx_tab)
    if (waves == 2) SPX_LAUNCH_SX(2);
    else if (waves == 8) SPX_LAUNCH_SX(8);
    else SPX_LAUNCH_SX(4);
#undef SPX_LAUNCH_SX
}

// the LDS a launch needs beyond the plain read-once kernel's: the pass headers (and the alignment in front)
size_t spmv_sx_header_bytes(uint32_t pass_stride)
{
    return 24u * ((size_t) pass_stride + 4u * MAX_WAVES_PER_BLOCK) + 32u;
}

void spmv_sx_allow_lds(size_t bytes)
{
    const int b = (int) bytes;
#define SPX_ATTR_SX(W) (void) hipFuncSetAttribute(reinterpret_cast<const void *>(&csx_spmv_sx_kernel<W, SX_PASSES_PER_ROUND>), hipFuncAttributeMaxDynamicSharedMemorySize, b)
    SPX_ATTR_SX(2); SPX_ATTR_SX(4); SPX_ATTR_SX(8);
#undef SPX_ATTR_SX
}

}  // namespace spx
