// spmv_xw_kernels.hip -- the general-path product with the unit windows of x staged in LDS and the unit
// passes software-pipelined (csx_spmv_xw_kernel; window plan: xwindows.hpp; launched by device_spmv in
// spmv_kernels.hip where the launch tuner found it faster than the plain kernel).
//
// Semantics as the plain kernel's: the reference's SpMV templates (src/templates/csx_spmv_tmpl.c:66-101,
// horiz_tmpl.c:20-37, diag_tmpl.c:20-35, block_row_tmpl.c, block_col_tmpl.c), every stored nonzero a(r,c)
// contributes alpha * a * x[c] to y[r].
#include "spmv_device.hpp"

#include <cstddef>

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
// one), and the wait for it is a wait for every value load in flight.
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

template <int W, int B>
struct XwStage {
    uint2 q[B];                                   // {offset of segment 0 in the unit windows, descriptor bits}
    uint32_t segl[B];                             // segments of the row-block in front of the lane's | active << 16
    double2 v2[B][W / 2 > 0 ? W / 2 : 1];
    double v1[B];
};

// The loads of B unit passes of width W: descriptors where the header holds none, values.  Everything the
// second half needs of the headers goes into the stage with them (an inline descriptor is copied, the lane's
// segment number and whether it is there at all are packed into one register): the headers themselves are
// dead once their loads are out, which is what lets the wavefront hold those of the next two rounds.
template <int W, int B>
__device__ __forceinline__ void xw_issue(const KernelArgs &a, const SpxRowBlock &rb, const PassWords (&ps)[B],
                                         XwStage<W, B> &S, int lane)
{
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const uint32_t nseg = ps[b].nseg();
        const bool active = (uint32_t) lane < nseg;
        const uint32_t l = active ? (uint32_t) lane : 0u;
        S.segl[b] = ps[b].seg0() + l + (active ? 0x10000u : 0u);
        // (always from the descriptor array, also where the header holds a copy: x comes from LDS once the
        // values are there, so the copy saves no round trip here, and a fixed number of loads per round
        // is what lets the compiler count them -- with a branch around this load it waited for the
        // previous round's values before requesting the next ones)
        const uint64_t mk = (ps[b].flags() & SPX_PASSF_INLINE) ? 0ull : ps[b].mask();
        const uint32_t rank = ps[b].rank0() + (active ? starts_upto(mk, lane) : 0u);
        S.q[b] = ld_stream(reinterpret_cast<const uint2 *>(a.descs + rb.desc_off + rank));
        const double *vals = a.values + rb.val_off + ps[b].val_off();
#pragma unroll
        for (int p = 0; p < W / 2; ++p)
            S.v2[b][p] = ld_stream(reinterpret_cast<const double2 *>(vals + (uint32_t) p * 2u * nseg + l * 2u));
        if (W & 1) S.v1[b] = ld_stream(vals + (uint32_t) (W / 2) * 2u * nseg + l);
    }
}

// ... and what follows once they have arrived: rows and window offsets, x from LDS, W FMAs, one LDS add
// (unit passes of a general stream: SpxPass::elem0 is 0)
template <int W, int B>
__device__ __forceinline__ void xw_finish(const XwStage<W, B> &S, double *tile, const double *xw)
{
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const uint32_t c0 = S.q[b].x, bits = S.q[b].y;
        const int s = (int) ((S.segl[b] - ((bits >> 9) & 8191u)) & 0xffffu);
        const uint32_t kind = (bits >> 22) & 7u;
        const int step = (int) (bits >> 25);
        const int drow = kind == SPX_KIND_BLOCK ? 1 : (kind >= SPX_KIND_VERT ? step : 0);
        const int dcol = (kind == SPX_KIND_HORIZ || kind == SPX_KIND_DIAG)
                             ? step : (kind == SPX_KIND_ADIAG ? -step : 0);
        const int row = (int) (bits & 511u) + s * drow;
        const double *xp = xw + (int) (c0 + (uint32_t) (s * dcol));      // (c0: an offset into the unit windows)
        double t = 0.0;
#ifdef SPX_XW_ABL_NOX          // (variant builds, tools/build_variant.sh: results wrong on purpose)
        const double one[8] = {1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0};
        xp = one;
#endif
#pragma unroll
        for (int p = 0; p < W / 2; ++p) {
            t = fma(S.v2[b][p].x, xp[2 * p], t);
            t = fma(S.v2[b][p].y, xp[2 * p + 1], t);
        }
        if (W & 1) t = fma(S.v1[b], xp[W - 1], t);
#ifdef SPX_XW_ABL_NOADD
        if (t == 1.2345e300 && row == 12345) atomicAdd(&tile[row], t);
#else
        if (S.segl[b] >> 16) atomicAdd(&tile[row], t);
#endif
    }
}

template <int W>
__device__ __forceinline__ void xw_single(const KernelArgs &a, const SpxRowBlock &rb, const PassWords &ps,
                                          double *tile, const double *xw, int lane)
{
    XwStage<W, 1> S;
    xw_issue<W, 1>(a, rb, {ps}, S, lane);
    xw_finish<W, 1>(S, tile, xw);
}

// one pass of any kind on its own
__device__ __forceinline__ void xw_one(const KernelArgs &a, const SpxRowBlock &rb, const PassWords &ps,
                                       double *tile, const double *win, const double *xw, int lane)
{
    if (ps.kind() == SPX_PASS_UNIT && (ps.flags() & SPX_PASSF_XLDS)) {
        switch (ps.width()) {            // wave-uniform
        case 1: xw_single<1>(a, rb, ps, tile, xw, lane); break;
        case 2: xw_single<2>(a, rb, ps, tile, xw, lane); break;
        case 3: xw_single<3>(a, rb, ps, tile, xw, lane); break;
        case 4: xw_single<4>(a, rb, ps, tile, xw, lane); break;
        case 5: xw_single<5>(a, rb, ps, tile, xw, lane); break;
        case 6: xw_single<6>(a, rb, ps, tile, xw, lane); break;
        case 7: xw_single<7>(a, rb, ps, tile, xw, lane); break;
        default: xw_single<8>(a, rb, ps, tile, xw, lane); break;
        }
    } else {
        run_pass(a, rb, ps.pass(), tile, win, lane);
    }
}

// Runs `n_rounds` >= 1 rounds t, t + 2 WAVES, ... -- pairs of unit passes of width W that read LDS, known
// from the row-block's table, not from their headers -- as a two-stage pipeline.  On entry (c0, c1) are the
// headers of round t and (n0, n1) those of the round after it; on return they are those of the first
// round that was not run, `t` its first pass.
// (A counted loop without a branch around any load: the compiler counts outstanding loads per path, and
// with the loads of the next round under a condition -- "is there a next round" read from its headers --
// it waited for every load before every use, which undid the pipeline.)
template <int W, int WAVES>
__device__ __forceinline__ void xw_run(const KernelArgs &a, const SpxRowBlock &rb, spx_const_words_t passes,
                                       int n_rounds, int &t, PassWords &c0, PassWords &c1, PassWords &n0, PassWords &n1,
                                       double *tile, const double *xw, int lane)
{
    // (the headers of the round after next are requested AFTER the loads of the next round went out and
    // are first looked at a whole round later: scalar loads complete out of order, so the wait in front of
    // their first use is a wait for everything scalar in flight)
#define SPX_XW_TAKE()                                                                             \
    do {                                                                                          \
        c0 = n0; c1 = n1;                                                                         \
        t += 2 * WAVES;                                                                           \
    } while (0)
#define SPX_XW_FETCH()                                                                            \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        n0 = load_pass(passes, t + 2 * WAVES); n1 = load_pass(passes, t + 3 * WAVES);             \
        __builtin_amdgcn_sched_barrier(0);                                                        \
    } while (0)
    XwStage<W, 2> A, B;
    xw_issue<W, 2>(a, rb, {c0, c1}, A, lane);
    int r = 1;
    for (; r + 1 < n_rounds; r += 2) {
        SPX_XW_TAKE();
        xw_issue<W, 2>(a, rb, {c0, c1}, B, lane);
        SPX_XW_FETCH();
        xw_finish<W, 2>(A, tile, xw);
        SPX_XW_TAKE();
        xw_issue<W, 2>(a, rb, {c0, c1}, A, lane);
        SPX_XW_FETCH();
        xw_finish<W, 2>(B, tile, xw);
    }
    if (r < n_rounds) {
        SPX_XW_TAKE();
        xw_issue<W, 2>(a, rb, {c0, c1}, B, lane);
        SPX_XW_FETCH();
        xw_finish<W, 2>(A, tile, xw);
        xw_finish<W, 2>(B, tile, xw);
    } else {
        xw_finish<W, 2>(A, tile, xw);
    }
    SPX_XW_TAKE();
    SPX_XW_FETCH();
#undef SPX_XW_TAKE
#undef SPX_XW_FETCH
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

    spx_const_words_t passes = (spx_const_words_t) (uintptr_t) (a.passes + (size_t) rb_idx * a.pass_stride);
    const SpxRowBlock rb = a.rbs[rb_idx];
    // the window table of the row-block: one entry per lane (a single load; the entries are handed
    // to the whole wavefront with v_readlane when their turn comes)
    const uint2 xw_entry = *reinterpret_cast<const uint2 *>(a.xw_tab + (size_t) rb_idx * XW_TAB + (lane & (XW_TAB - 1)));
    PassWords c0 = load_pass(passes, wave), c1 = load_pass(passes, wave + WAVES);       // (the table is padded)
    PassWords n0 = load_pass(passes, wave + 2 * WAVES), n1 = load_pass(passes, wave + 3 * WAVES);
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
    uint32_t odd_base = 0, odd_at = 0xffffffffu;
#ifndef SPX_XW_ABL_NOSTAGE
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
#endif
    if (odd_at != 0xffffffffu && threadIdx.x == 0) xw[odd_at] = a.x[odd_base];
    __syncthreads();

    // where the pipeline may run: per width 1..4 the passes [lo, hi) of the row-block that are unit passes of
    // that width reading LDS (the first entries of the table)
    const uint32_t range12_lo = (uint32_t) __builtin_amdgcn_readlane((int) xw_entry.x, 0);
    const uint32_t range12_hi = (uint32_t) __builtin_amdgcn_readlane((int) xw_entry.y, 0);
    const uint32_t range34_lo = (uint32_t) __builtin_amdgcn_readlane((int) xw_entry.x, 1);
    const uint32_t range34_hi = (uint32_t) __builtin_amdgcn_readlane((int) xw_entry.y, 1);
    const int n_pass = rb.n_pass;
    int t = wave;
    while (t < n_pass) {
        // rounds from t on whose two passes both lie inside one of the ranges
        int n_rounds = 0, width = 0;
        {
            const uint32_t rg[4] = {range12_lo, range12_hi, range34_lo, range34_hi};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int lo = (int) (rg[w] & 0xffffu), hi = (int) (rg[w] >> 16);
                if (t >= lo && t + WAVES < hi) {
                    n_rounds = (hi - 1 - WAVES - t) / (2 * WAVES) + 1;
                    width = w + 1;
                }
            }
        }
        if (n_rounds > 0) {
            switch (width) {        // wave-uniform
            case 1: xw_run<1, WAVES>(a, rb, passes, n_rounds, t, c0, c1, n0, n1, tile, xw, lane); break;
            case 2: xw_run<2, WAVES>(a, rb, passes, n_rounds, t, c0, c1, n0, n1, tile, xw, lane); break;
            case 3: xw_run<3, WAVES>(a, rb, passes, n_rounds, t, c0, c1, n0, n1, tile, xw, lane); break;
            default: xw_run<4, WAVES>(a, rb, passes, n_rounds, t, c0, c1, n0, n1, tile, xw, lane); break;
            }
            continue;
        }
        const bool two = t + WAVES < n_pass;
        xw_one(a, rb, c0, tile, win, xw, lane);
        if (two) xw_one(a, rb, c1, tile, win, xw, lane);
        c0 = n0; c1 = n1;
        n0 = load_pass(passes, t + 4 * WAVES); n1 = load_pass(passes, t + 5 * WAVES);
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

// the same product with the unit windows of x staged in LDS and the unit passes pipelined (spmv_body_xw)
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES)
void csx_spmv_xw_kernel(SPX_KERNEL_PARAMS, const XwEntry *xw_tab_)
{
    SPX_KERNEL_ARGS(a);
    a.xw_tab = xw_tab_;
    extern __shared__ double lds_dyn[];      // y tile, the leftovers' x window, the unit windows
    spmv_body_xw<WAVES>(a, xcd_split, lds_dyn);
}

void launch_spmv_xw(int waves, unsigned blocks, size_t lds_bytes, void *stream_, const KernelArgs &a,
                    const XcdSplit &xs)
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
