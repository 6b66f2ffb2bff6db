// spmv_xwp_kernels.hip -- the general-path product with persistent workgroups: unit windows of x in LDS, double
// buffered across row-blocks; every wavefront runs ONE precompiled list of rounds from its first row-block to
// its last as a software pipeline that never drains in between (csx_spmv_xwp_kernel; plan: xwindows.hpp,
// plan_persistent_rounds; launched by device_spmv in spmv_kernels.hip where the launch tuner found it fastest).
//
// Semantics as the plain kernel's: the reference's SpMV templates (src/templates/csx_spmv_tmpl.c:66-101,
// horiz_tmpl.c:20-37, diag_tmpl.c:20-35, block_row_tmpl.c, block_col_tmpl.c), every stored nonzero a(r,c)
// contributes alpha * a * x[c] to y[r].
//
// A workgroup's LDS holds two regions, each a y tile and the unit windows of one row-block.  Row-block k of
// the workgroup's list uses region k & 1.  When a wavefront has finished its last round of row-block k
// (XWP_LAST) it runs the row-block's END:
//     1. its passes outside the pipeline (leftovers, wide units), if any, on their own;
//     2. the window pieces of row-block k + 1 it has been holding in registers go to region (k + 1) & 1
//        (free since the end of row-block k - 1);
//     3. barrier: all sums of row-block k are in its tile, all windows of row-block k + 1 in LDS;
//     4. y <- alpha * tile + beta * y for the rows of row-block k, the tile is cleared for row-block k + 2;
//     5. the loads of row-block k + 2's window pieces go out (registers until the next end).
// One barrier per row-block, and nothing in it waits for memory that was not requested a row-block earlier.
// The wavefront's loads for the rounds that follow are in flight the whole time.
#include "spmv_device.hpp"

#include <cstddef>
#include <cstdio>
#include <cstdlib>

namespace spx {

struct XwpArgs {
    const SpxRowBlock *rbs;
    const SpxPass *passes;          // (the unit-window copy: SPX_PASSF_XLDS, translated inline descriptors)
    const double *values;
    const SpxUnitDesc *descs;       // (the unit-window copy, followed by the descriptors of the leftover halves: XwPlan::gdesc)
    const uint8_t *cidx;
    const uint16_t *segrows;
    const double *x;
    double *y;
    const XwEntry *xw_tab;
    const XwpRound *rounds;
    const uint64_t *stream_off;
    const uint32_t *stream_len;
    double alpha, beta;
    uint32_t wgs_per_xcd;
    uint32_t pass_stride;
    uint32_t region;                // doubles per LDS region
    uint32_t tile_rows;             // ... of which the y tile (even)
};

typedef const __attribute__((address_space(4))) uint32_t *xwp_words_t;

// One stage: two unit passes of any width 1..4 (three loads each, as in spmv_xw_kernels.hip), and the record
// of the round that goes into this stage next, one dword per lane (lanes 0-15)
struct XwpStage {
    uint2 q[2];
    uint32_t segl[2], width[2], row0[2];
    spx_d2u_t va[2], vb[2];
    uint32_t flags;                 // XWP_* of the round (wave-uniform)
    uint32_t next, next2;           // lane l < 16: dword l of the records of the next two rounds of this stage
};

__device__ __forceinline__ uint32_t xwp_word(uint32_t v, int k)
{
    return (uint32_t) __builtin_amdgcn_readlane((int) v, k);
}

// the loads of the round whose record the stage holds; then the load of the record 2 D rounds on
__device__ __forceinline__ void xwp_issue(const XwpArgs &a, XwpStage &S, const XwpRound *next_record, int lane)
{
    const uint32_t rec = S.next;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const uint64_t val = (uint64_t) xwp_word(rec, 6 * p) | ((uint64_t) xwp_word(rec, 6 * p + 1) << 32);
        const uint32_t desc = xwp_word(rec, 6 * p + 2);
        const uint64_t mask = (uint64_t) xwp_word(rec, 6 * p + 3) | ((uint64_t) xwp_word(rec, 6 * p + 4) << 32);
        const uint32_t geom = xwp_word(rec, 6 * p + 5);
        const uint32_t seg0 = geom & 0xffffu, nseg = (geom >> 16) & 0xffu, W = geom >> 24;
        const bool active = (uint32_t) lane < nseg;
        const uint32_t l = active ? (uint32_t) lane : 0u;
        S.segl[p] = seg0 + l + (active ? 0x10000u : 0u);
        S.width[p] = W;
        S.row0[p] = xwp_word(rec, 12 + p);
        // (half of a leftover pass: a descriptor of its own per lane)
        const bool gather = W >= XWP_WIDTH_GATHER2;
        uint32_t rank = desc;
        if (gather) rank += l;
        else if (mask != 0ull) rank += active ? starts_upto(mask, lane) : 0u;
        // (the descriptors of the leftover halves follow the unit descriptors in ONE array: the record holds
        // the index either way -- a choice between two pointers here put both into scratch memory)
        S.q[p] = ld_stream(reinterpret_cast<const uint2 *>(a.descs + rank));
        const double *vals = a.values + val;
        const uint32_t off_a = (W == 1u || W == XWP_WIDTH_GATHER1) ? l : 2u * l;
        const uint32_t off_b = W == 3u ? 2u * nseg + l : (W == 4u ? 2u * nseg + 2u * l : off_a);
        S.va[p] = *reinterpret_cast<const spx_d2u_t *>(vals + off_a);
        S.vb[p] = *reinterpret_cast<const spx_d2u_t *>(vals + off_b);
    }
    S.flags = xwp_word(rec, 14);
    // (records travel two turns of the pipeline ahead: every other one is the first touch of its 128-byte line
    // and comes all the way from HBM)
    S.next = S.next2;
    S.next2 = reinterpret_cast<const uint32_t *>(next_record)[lane & 15];
}

template <int W>
__device__ __forceinline__ void xwp_finish_pass(uint2 q, uint32_t segl, uint32_t row0, spx_d2u_t va, spx_d2u_t vb,
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
    const double *xp = xw + (int) (c0 + (uint32_t) (s * dcol));
    double t = va.x * xp[0];
    if (W >= 2) t = fma(va.y, xp[1], t);
    if (W >= 3) t = fma(vb.x, xp[2], t);
    if (W >= 4) t = fma(vb.y, xp[3], t);
    if (segl >> 16) atomicAdd(&tile[row], t);
}

// half of a leftover pass: up to two nonzeros of one row per lane, x from the windows
__device__ __forceinline__ void xwp_finish_gather(uint2 q, uint32_t segl, spx_d2u_t va, double *tile, const double *xw)
{
    const uint32_t valid = q.y >> 16;
    const double x0 = xw[q.x & 0xffffu], x1 = xw[q.x >> 16];
    double t = valid >= 1u ? va.x * x0 : 0.0;
    if (valid >= 2u) t = fma(va.y, x1, t);
    if (segl >> 16) atomicAdd(&tile[q.y & 0xffffu], t);
}

__device__ __forceinline__ void xwp_finish(const XwpStage &S, double *tile, const double *xw)
{
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        switch (S.width[p]) {          // wave-uniform
        case XWP_WIDTH_GATHER2: case XWP_WIDTH_GATHER1: xwp_finish_gather(S.q[p], S.segl[p], S.va[p], tile, xw); break;
        case 1: xwp_finish_pass<1>(S.q[p], S.segl[p], S.row0[p], S.va[p], S.vb[p], tile, xw); break;
        case 2: xwp_finish_pass<2>(S.q[p], S.segl[p], S.row0[p], S.va[p], S.vb[p], tile, xw); break;
        case 3: xwp_finish_pass<3>(S.q[p], S.segl[p], S.row0[p], S.va[p], S.vb[p], tile, xw); break;
        default: xwp_finish_pass<4>(S.q[p], S.segl[p], S.row0[p], S.va[p], S.vb[p], tile, xw); break;
        }
    }
}

// a unit pass of width 5..8 that reads LDS, on its own
template <int W>
__device__ __forceinline__ void xwp_wide(const XwpArgs &a, const SpxRowBlock &rb, const SpxPass &ps, double *tile,
                                         const double *xw, int lane)
{
    const uint32_t nseg = ps.nseg;
    const bool active = (uint32_t) lane < nseg;
    const uint32_t l = active ? (uint32_t) lane : 0u;
    const uint64_t mk = (ps.flags & SPX_PASSF_INLINE) ? 0ull : ps.mask;
    const uint32_t rank = ps.rank0 + (active ? starts_upto(mk, lane) : 0u);
    const uint2 q = ld_stream(reinterpret_cast<const uint2 *>(a.descs + rb.desc_off + rank));
    const double *vals = a.values + rb.val_off + ps.val_off;
    double2 v2[W / 2];
    double v1 = 0.0;
#pragma unroll
    for (int p = 0; p < W / 2; ++p)
        v2[p] = ld_stream(reinterpret_cast<const double2 *>(vals + (uint32_t) p * 2u * nseg + l * 2u));
    if (W & 1) v1 = ld_stream(vals + (uint32_t) (W / 2) * 2u * nseg + l);
    const uint32_t bits = q.y;
    const int s = (int) ((ps.seg0 + l - ((bits >> 9) & 8191u)) & 0xffffu);
    const uint32_t kind = (bits >> 22) & 7u;
    const int step = (int) (bits >> 25);
    const int drow = kind == SPX_KIND_BLOCK ? 1 : (kind >= SPX_KIND_VERT ? step : 0);
    const int dcol = (kind == SPX_KIND_HORIZ || kind == SPX_KIND_DIAG) ? step : (kind == SPX_KIND_ADIAG ? -step : 0);
    const int row = (int) (ps.elem0 + (bits & 511u)) + s * drow;
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

// one pass outside the pipeline: a leftover pass or a unit pass that gathers through L2 (the plain kernel's
// code), a narrow unit pass that reads LDS (a stage of its own), a wide one
__device__ __forceinline__ void xwp_one(const XwpArgs &a, const SpxRowBlock &rb, const SpxPass &ps, double *tile,
                                        const double *xw, int lane)
{
    if (ps.kind == SPX_PASS_UNIT && (ps.flags & SPX_PASSF_XLDS)) {
        switch (ps.width) {            // wave-uniform
        case 1: case 2: case 3: case 4: {
            const uint32_t nseg = ps.nseg, W = ps.width;
            const bool active = (uint32_t) lane < nseg;
            const uint32_t l = active ? (uint32_t) lane : 0u;
            const uint64_t mk = (ps.flags & SPX_PASSF_INLINE) ? 0ull : ps.mask;
            const uint32_t rank = ps.rank0 + (active ? starts_upto(mk, lane) : 0u);
            const uint2 q = ld_stream(reinterpret_cast<const uint2 *>(a.descs + rb.desc_off + rank));
            const double *vals = a.values + rb.val_off + ps.val_off;
            const uint32_t off_a = W == 1u ? l : 2u * l;
            const uint32_t off_b = W == 3u ? 2u * nseg + l : (W == 4u ? 2u * nseg + 2u * l : off_a);
            const spx_d2u_t va = *reinterpret_cast<const spx_d2u_t *>(vals + off_a);
            const spx_d2u_t vb = *reinterpret_cast<const spx_d2u_t *>(vals + off_b);
            const uint32_t segl = ps.seg0 + l + (active ? 0x10000u : 0u);
            if (W == 1) xwp_finish_pass<1>(q, segl, ps.elem0, va, vb, tile, xw);
            else if (W == 2) xwp_finish_pass<2>(q, segl, ps.elem0, va, vb, tile, xw);
            else if (W == 3) xwp_finish_pass<3>(q, segl, ps.elem0, va, vb, tile, xw);
            else xwp_finish_pass<4>(q, segl, ps.elem0, va, vb, tile, xw);
            break;
        }
        case 5: xwp_wide<5>(a, rb, ps, tile, xw, lane); break;
        case 6: xwp_wide<6>(a, rb, ps, tile, xw, lane); break;
        case 7: xwp_wide<7>(a, rb, ps, tile, xw, lane); break;
        default: xwp_wide<8>(a, rb, ps, tile, xw, lane); break;
        }
    } else {
        KernelArgs ka;
        ka.rbs = a.rbs; ka.passes = a.passes; ka.values = a.values; ka.descs = a.descs; ka.cidx = a.cidx;
        ka.segrows = a.segrows; ka.x = a.x; ka.y = a.y; ka.alpha = a.alpha; ka.beta = a.beta;
        run_pass(ka, rb, ps, tile, nullptr, lane);
    }
}

// The loader wavefront's part of a row-block: its unit windows straight into LDS (global_load_lds_dwordx4,
// pieces of 128 doubles, 16 bytes per lane; a wavefront that keeps no loaded values in registers may use
// LDS DMA freely -- for the others the compiler would wait for everything at every use).  `tabv`: the
// row-block's window table, one entry per lane.
__device__ __forceinline__ void xwp_stage_windows(const double *x, uint2 tabv, double *xw, int lane)
{
#pragma unroll 1
    for (uint32_t k = 0; k < XW_MAX; ++k) {
        const uint32_t base = (uint32_t) __builtin_amdgcn_readlane((int) tabv.x, (int) (XW_RANGES + k));
        const uint32_t off_len = (uint32_t) __builtin_amdgcn_readlane((int) tabv.y, (int) (XW_RANGES + k));
        const uint32_t len = off_len >> 16, off = off_len & 0xffffu;
        if (len == 0) break;
        const double *src = x + base;
        for (uint32_t c = 0; c < len; c += 128u) {
            const uint32_t i = c + 2u * (uint32_t) lane;
            if (i < len)
                __builtin_amdgcn_global_load_lds(src + i, (__attribute__((address_space(3))) void *) (xw + off + c), 16, 0, 0);
        }
    }
}

// (the pointers are kernel parameters of their own, not members of a struct passed by value: only so does
// the compiler know that they point to global memory -- through a struct they became flat loads, which count
// against both wait counters and made every wait a wait for everything)
struct XwpScalars {
    double alpha, beta;
    uint32_t first[9];
    uint32_t wgs_per_xcd, pass_stride, region, tile_rows;
};

// GEN: the list holds row-blocks with passes outside the pipeline (their code costs fifty registers: a build
// without it for the streams that have none)
template <int WAVES, int D, bool GEN>
__global__ __launch_bounds__(64 * (WAVES + 1))
void csx_spmv_xwp_kernel(const SpxRowBlock *rbs_, const SpxPass *passes_, const double *values_, const SpxUnitDesc *descs_,
                         const uint8_t *cidx_, const uint16_t *segrows_, const double *x_, double *y_,
                         const XwEntry *xw_tab_, const XwpRound *rounds_, const uint64_t *stream_off_,
                         const uint32_t *stream_len_, const XwpScalars sc)
{
    XwpArgs a;
    a.rbs = rbs_; a.passes = passes_; a.values = values_; a.descs = descs_; a.cidx = cidx_; a.segrows = segrows_;
    a.x = x_; a.y = y_; a.xw_tab = xw_tab_; a.rounds = rounds_; a.stream_off = stream_off_;
    a.stream_len = stream_len_; a.alpha = sc.alpha; a.beta = sc.beta;
    a.wgs_per_xcd = sc.wgs_per_xcd; a.pass_stride = sc.pass_stride; a.region = sc.region; a.tile_rows = sc.tile_rows;
    constexpr int BLOCK = 64 * (WAVES + 1);          // WAVES wavefronts run passes, one more stages and writes
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t xcd = blockIdx.x & 7u, g = blockIdx.x >> 3;
    const uint32_t rb_lo = sc.first[xcd] + g, rb_hi = sc.first[xcd + 1u], G = a.wgs_per_xcd;
    if (rb_lo >= rb_hi) return;
    const uint32_t n_blocks = (rb_hi - rb_lo + G - 1u) / G;          // row-blocks of this workgroup

    double *const tile0 = lds, *const tile1 = lds + a.region;
    double *const xw0 = tile0 + a.tile_rows, *const xw1 = tile1 + a.tile_rows;
    for (uint32_t i = tid; i < a.tile_rows; i += BLOCK) {
        tile0[i] = 0.0;
        tile1[i] = 0.0;
    }

    if (wave == WAVES) {
        // ---- the loader: windows of row-block k + 1 into the region row-block k - 1 has left, the barrier
        // that ends row-block k, then its rows of y and a cleared tile.  Window tables travel two row-blocks
        // ahead (one entry per lane; the spare entry holds the row-block's first row and row count).
        auto table = [&](uint32_t kk) {
            const uint32_t rb = kk < n_blocks ? rb_lo + kk * G : rb_lo;
            return *reinterpret_cast<const uint2 *>(a.xw_tab + (size_t) rb * XW_TAB + (lane & (XW_TAB - 1)));
        };
        uint2 tab_a = table(0), tab_b = table(1), tab_c = table(2);      // row-blocks k, k + 1, k + 2
        xwp_stage_windows(a.x, tab_a, xw0, lane);
        if (n_blocks > 1u) xwp_stage_windows(a.x, tab_b, xw1, lane);
        __syncthreads();                                                 // (waits for the LDS DMA as well)
        for (uint32_t k = 0; k < n_blocks; ++k) {
            // (the windows of row-block k + 1 were requested an iteration ago: __syncthreads waits for them)
            __syncthreads();                                             // the end of row-block k
            double *const tile = (k & 1u) ? tile1 : tile0;
            const uint32_t row0 = (uint32_t) __builtin_amdgcn_readlane((int) tab_a.x, 1);
            const uint32_t n_rows = (uint32_t) __builtin_amdgcn_readlane((int) tab_a.y, 1);
            for (uint32_t i = lane; i < n_rows; i += 64u) {
                const size_t gr = (size_t) row0 + i;
                double t = a.alpha * tile[i];
                if (a.beta != 0.0) t += a.beta * a.y[gr];
                a.y[gr] = t;
                tile[i] = 0.0;
            }
            tab_a = tab_b;
            tab_b = tab_c;
            if (k + 2u < n_blocks) xwp_stage_windows(a.x, tab_b, (k & 1u) ? xw1 : xw0, lane);
            tab_c = table(k + 3u);
        }
        return;
    }

    // ---- the wavefronts that run the passes: one list of rounds each, D of them in flight
    const size_t stream = (size_t) blockIdx.x * WAVES + (size_t) wave;
    const XwpRound *rec = a.rounds + a.stream_off[stream];
    const uint32_t n_rounds = a.stream_len[stream];
    XwpStage st[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        st[d].next = reinterpret_cast<const uint32_t *>(rec + d)[lane & 15];
        st[d].next2 = reinterpret_cast<const uint32_t *>(rec + D + d)[lane & 15];
    }
#pragma unroll
    for (int d = 0; d < D; ++d) xwp_issue(a, st[d], rec + 2 * D + d, lane);
    // (the first barrier: tiles cleared, the first windows in LDS.  Raw barriers from here on: a __syncthreads
    // would wait for every load in flight, and the point of the list is that they stay in flight)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    uint32_t k = 0;                  // row-block of the list that the round being finished belongs to
    for (uint32_t r = 0; r < n_rounds; r += D) {
#pragma clang loop unroll(full)
        for (int d = 0; d < D; ++d) {
            // (rounds behind the list's end are empty: no lanes, no flags)
            xwp_finish(st[d], (k & 1u) ? tile1 : tile0, (k & 1u) ? xw1 : xw0);
            if (st[d].flags & XWP_LAST) {
                // ---- the end of row-block k: passes outside the pipeline, if this wavefront has any; then
                // all sums are in the tile -- the loader takes it from there
                if (GEN && (st[d].flags & XWP_GENERIC)) {
                    const uint32_t rb_idx = rb_lo + k * G;
                    const SpxRowBlock rb = a.rbs[rb_idx];
                    const uint32_t range = a.xw_tab[(size_t) rb_idx * XW_TAB].base, lo = range & 0xffffu, hi = range >> 16;
                    const SpxPass *ps = a.passes + (size_t) rb_idx * a.pass_stride;
                    for (uint32_t t = (uint32_t) wave; t < rb.n_pass; t += WAVES)
                        if ((t < lo || t >= hi) && !(ps[t].kind == SPX_PASS_GATHER && (ps[t].flags & SPX_PASSF_XLDS)))
                            xwp_one(a, rb, ps[t], (k & 1u) ? tile1 : tile0, (k & 1u) ? xw1 : xw0, lane);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                ++k;
            }
            xwp_issue(a, st[d], rec + r + d + 3 * D, lane);
        }
    }
}

void launch_spmv_xwp(int waves, int depth, bool generic, unsigned blocks, size_t lds_bytes, void *stream_, const XwpArgs &a,
                     const uint32_t first[9])
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    XwpScalars sc;
    sc.alpha = a.alpha; sc.beta = a.beta;
    for (int k = 0; k < 9; ++k) sc.first[k] = first[k];
    sc.wgs_per_xcd = a.wgs_per_xcd; sc.pass_stride = a.pass_stride; sc.region = a.region; sc.tile_rows = a.tile_rows;
#define SPX_LAUNCH_XWP(W, DD, GG)                                                                        \
    hipLaunchKernelGGL((csx_spmv_xwp_kernel<W, DD, GG>), dim3(blocks), dim3(64 * (W + 1)), lds_bytes, stream, a.rbs,   \
                       a.passes, a.values, a.descs, a.cidx, a.segrows, a.x, a.y, a.xw_tab, a.rounds,          \
                       a.stream_off, a.stream_len, sc)
#define SPX_LAUNCH_XWP_D(W, GG)                                                                          \
    do {                                                                                                 \
        if (depth >= 4) SPX_LAUNCH_XWP(W, 4, GG);                                                        \
        else if (depth == 3) SPX_LAUNCH_XWP(W, 3, GG);                                                   \
        else SPX_LAUNCH_XWP(W, 2, GG);                                                                   \
    } while (0)
#define SPX_LAUNCH_XWP_W(GG)                                                                             \
    do {                                                                                                 \
        if (waves == 8) SPX_LAUNCH_XWP_D(8, GG);                                                         \
        else if (waves == 7) SPX_LAUNCH_XWP_D(7, GG);                                                    \
        else if (waves == 3) SPX_LAUNCH_XWP_D(3, GG);                                                    \
        else SPX_LAUNCH_XWP_D(4, GG);                                                                    \
    } while (0)
    if (generic) SPX_LAUNCH_XWP_W(true);
    else SPX_LAUNCH_XWP_W(false);
#undef SPX_LAUNCH_XWP_W
#undef SPX_LAUNCH_XWP_D
#undef SPX_LAUNCH_XWP
}

void xwp_launch(int waves, int depth, unsigned blocks, size_t lds_bytes, void *stream, const SpxRowBlock *rbs,
                const SpxPass *passes, const double *values, const SpxUnitDesc *descs, const uint8_t *cidx,
                const uint16_t *segrows, const double *x, double *y, const XwEntry *xw_tab,
                const XwpRound *rounds,
                const uint64_t *stream_off, const uint32_t *stream_len, double alpha, double beta,
                const uint32_t first[9], uint32_t wgs_per_xcd, uint32_t pass_stride, uint32_t region, uint32_t tile_rows,
                bool generic)
{
    XwpArgs a;
    a.rbs = rbs; a.passes = passes; a.values = values; a.descs = descs; a.cidx = cidx; a.segrows = segrows;
    a.x = x; a.y = y; a.xw_tab = xw_tab; a.rounds = rounds; a.stream_off = stream_off; a.stream_len = stream_len;
    a.alpha = alpha; a.beta = beta;
    a.wgs_per_xcd = wgs_per_xcd; a.pass_stride = pass_stride; a.region = region; a.tile_rows = tile_rows;
    launch_spmv_xwp(waves, depth, generic, blocks, lds_bytes, stream, a, first);
}

void spmv_xwp_allow_lds(size_t bytes)
{
    const int b = (int) bytes;
#define SPX_XWP_ATTR(W, DD)                                                                                          \
    (void) hipFuncSetAttribute(reinterpret_cast<const void *>(&csx_spmv_xwp_kernel<W, DD, true>), hipFuncAttributeMaxDynamicSharedMemorySize, b);  \
    (void) hipFuncSetAttribute(reinterpret_cast<const void *>(&csx_spmv_xwp_kernel<W, DD, false>), hipFuncAttributeMaxDynamicSharedMemorySize, b)
    SPX_XWP_ATTR(3, 2); SPX_XWP_ATTR(3, 3); SPX_XWP_ATTR(3, 4);
    SPX_XWP_ATTR(4, 2); SPX_XWP_ATTR(4, 3); SPX_XWP_ATTR(4, 4);
    SPX_XWP_ATTR(7, 2); SPX_XWP_ATTR(7, 3); SPX_XWP_ATTR(7, 4);
    SPX_XWP_ATTR(8, 2); SPX_XWP_ATTR(8, 3); SPX_XWP_ATTR(8, 4);
#undef SPX_XWP_ATTR
}

}  // namespace spx
