// spmv_sym_device.hpp -- device code of the symmetric passes whose values are read ONCE and used twice
// (SPX_PASS_SYMTILE, SPX_PASS_SYMSEG; gpu_format.h), shared by the kernels of spmv_kernels.hip and the
// pipelined read-once kernel of spmv_sx_kernels.hip.
//
// Semantics restated from the reference's symmetric SpMV template (src/templates/csx_sym_spmv_tmpl.c:60-106
// and the *_sym_tmpl.c unit bodies): a stored a(r,c) of the strictly lower triangle contributes
// alpha * a * x[c] to y[r] and alpha * a * x[r] to y[c] (`cur[c] += x[r] * v * alpha`, :92-95).
#pragma once

#include "spmv_device.hpp"
#include "spx_abl.hpp"

namespace spx {

// lane ^ 1, ^ 2, ^ 4 inside groups of eight lanes as DPP moves (VALU) instead of
// ds_bpermute (__shfl_xor goes through the LDS crossbar): quad_perm for 1 and 2,
// row_half_mirror followed by a reversed quad for 4 (lane i <- 7-i <- (7-i)^3 = i^4).
template <int DPP_CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), DPP_CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), DPP_CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double xchg1(double v)
{
    return dpp_mov_f64<0xB1>(v);                       // quad_perm [1,0,3,2]
}
__device__ __forceinline__ double xchg2(double v)
{
    return dpp_mov_f64<0x4E>(v);                       // quad_perm [2,3,0,1]
}
__device__ __forceinline__ double xchg4(double v)
{
    return dpp_mov_f64<0x1B>(dpp_mov_f64<0x141>(v));   // row_half_mirror, then quad_perm [3,2,1,0]
}

// A pass of symmetric tiles (SPX_PASS_SYMTILE): lanes 8t..8t+7 hold the rows of
// the dense 8x8 tile t of the stored lower triangle.  Each value is read once
// and used twice: a(r,c)*x[c] summed along the lane's row goes to the y tile,
// a(r,c)*x[r] summed over the tile's eight lanes goes to the slot of column c.
// The column sums are formed in registers by a three-step exchange within the
// eight lanes (4 + 2 + 1 values travel), so that each lane ends up with ONE
// column and the LDS adds of a tile hit eight different addresses -- lanes
// that add to the same address are serialised at ~3 clocks each.
// the lane's tile descriptor of a tile pass: {col0, row0 | slot << 9}
__device__ __forceinline__ uint2 symtile_desc(const KernelArgs &a, const SpxRowBlock &rb, const SpxPass &ps, int lane)
{
    const uint32_t l = (uint32_t) lane < ps.nseg ? (uint32_t) lane : 0u;
    return *reinterpret_cast<const uint2 *>(a.descs + rb.desc_off + ps.rank0 + (l >> 3));
}

// (`q`: the lane's tile descriptor, fetched by the caller -- a wavefront that runs several tile passes in a row
// asks for the next pass' descriptors together with the values of the current one: one memory round trip per
// pass instead of two, symtile_run below)
__device__ __forceinline__ void symtile_pass(const KernelArgs &a, const SpxRowBlock &rb,
                                             const SpxPass &ps, const uint2 q, double *slots, double *tile,
                                             int lane)
{
    const uint32_t nseg = ps.nseg;
    const bool active = (uint32_t) lane < nseg;
    const uint32_t l = active ? (uint32_t) lane : 0u;
    const double *vals = a.values + rb.val_off + ps.val_off;
    double2 v2[4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
        v2[p] = ld_stream(reinterpret_cast<const double2 *>(vals + (uint32_t) p * 2u * nseg + l * 2u));
    const int i = (int) (l & 7u);
    const int row = (int) (ps.elem0 + (q.y & 511u)) + i;
    const uint32_t slot = q.y >> 9;
    const double xr = a.x[rb.row0 + (uint32_t) row];
    const double *xp = a.x + q.x;
    double v[8], t = 0.0, p8[8];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        v[2 * p] = v2[p].x;
        v[2 * p + 1] = v2[p].y;
    }
    // (tiles start on columns that are multiples of eight: where x itself is 16-byte
    // aligned the eight x values of the tile come as four 16-byte loads)
    double xc[8];
    if ((reinterpret_cast<uintptr_t>(a.x) & 15u) == 0) {
        const double2 *xp2 = reinterpret_cast<const double2 *>(xp);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const double2 xx = xp2[p];
            xc[2 * p] = xx.x;
            xc[2 * p + 1] = xx.y;
        }
    } else {
#pragma unroll
        for (int w = 0; w < 8; ++w) xc[w] = xp[w];
    }
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        t = fma(v[w], xc[w], t);
        // (no masking of idle lanes here: a pass holds whole tiles -- nseg is a multiple of eight -- and the exchange
        // below stays inside a tile's eight lanes, so what an idle tile's lanes carry never reaches a live one; they
        // are kept from adding at the end.  Sixteen conditional moves and their registers less.)
        p8[w] = v[w] * xr;
    }
    // exchange with lane^4: lanes 0-3 collect columns 0-3, lanes 4-7 columns 4-7
    double p4[4];
    {
        const bool hi = (i & 4) != 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const double send = hi ? p8[w] : p8[w + 4];
            const double keep = hi ? p8[w + 4] : p8[w];
            p4[w] = keep + xchg4(send);
        }
    }
    // lane^2: lanes with bit 1 clear keep the lower two of their four columns
    double p2[2];
    {
        const bool hi = (i & 2) != 0;
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const double send = hi ? p4[w] : p4[w + 2];
            const double keep = hi ? p4[w + 2] : p4[w];
            p2[w] = keep + xchg2(send);
        }
    }
    // lane^1: one column each -- lane i of the tile holds column i
    double cs;
    {
        const bool hi = (i & 1) != 0;
        const double send = hi ? p2[0] : p2[1];
        const double keep = hi ? p2[1] : p2[0];
        cs = keep + xchg1(send);
    }
    if (active) {
        atomicAdd(&tile[row], t);
        atomicAdd(&slots[slot + (uint32_t) i], cs);
    }
}

__device__ __forceinline__ void symtile_pass(const KernelArgs &a, const SpxRowBlock &rb, const SpxPass &ps,
                                             double *slots, double *tile, int lane)
{
    symtile_pass(a, rb, ps, symtile_desc(a, rb, ps, lane), slots, tile, lane);
}

// The tile passes at the head of a wavefront's list (pass t, t + WAVES, ... while they are tile passes; the
// emitter puts a row-block's tiles first), one memory round trip each: a tile's x addresses come from its
// descriptor, and fetched inside the pass the descriptor costs a round trip of its own in front of the x loads
// (a chain header -> descriptor -> x; syn-nd24k: sixteen passes of a row-block on four wavefronts, every one
// of them two dependent round trips out of the Infinity Cache).  Here the descriptors of the NEXT pass are
// requested in front of the loads of the current one, the pass header a pass further ahead still.
// `p0` / `p1`: the headers of the passes t and t + WAVES, already loaded.  On return t is the wavefront's next
// pass (not a tile pass, or past the end) and p0 / p1 are its header and the one WAVES behind it.
template <int WAVES>
__device__ __forceinline__ void symtile_run(const KernelArgs &a, const SpxRowBlock &rb, const SpxPass *passes, int n_pass,
                                            int &t, SpxPass &p0, SpxPass &p1, double *slots, double *tile, int lane)
{
    if (t >= n_pass || p0.kind != SPX_PASS_SYMTILE) return;
    uint2 q = symtile_desc(a, rb, p0, lane);
    for (;;) {
        const bool more = t + WAVES < n_pass && p1.kind == SPX_PASS_SYMTILE;         // (wave-uniform)
        const SpxPass p2 = passes[t + 2 * WAVES];                                    // (the table is padded)
        const uint2 qn = symtile_desc(a, rb, more ? p1 : p0, lane);
        symtile_pass(a, rb, p0, q, slots, tile, lane);
        t += WAVES;
        p0 = p1;
        p1 = p2;
        q = qn;
        if (!more) break;
    }
}

// A pass of read-once row segments of a symmetric matrix (SPX_PASS_SYMSEG): a unit pass
// whose lanes, besides the row sum a(r, c..c+W-1) . x[c..], add the W transposed products
// a(r, c+w) * x[r] to the slots of their columns (consecutive slots, consecutive LDS
// addresses; lanes of neighbouring rows mostly hit different ones) -- every value is read
// once and used twice.  A segment without slots adds straight to y (global atomics; the
// kernel's hand-over is atomic anyway).
template <int W, int B>
__device__ __forceinline__ void symseg_passes(const KernelArgs &a, const SpxRowBlock &rb,
                                              const SpxPass (&ps)[B], double *slots, double *tile, int lane)
{
    // (B passes of the same width at once, stage by stage like the unit passes: all descriptors,
    // then all values, then x -- the pass is a chain of dependent loads, and with three values
    // per lane one pass alone keeps too little in flight: the ablation build that hands nothing
    // over still took 0.96 of the full kernel's 1.07 ms on the bench matrix)
    bool active[B];
    uint32_t l[B], nseg[B], slot0[B];
    uint2 q[B];
#pragma unroll
    for (int b = 0; b < B; ++b) {
        nseg[b] = ps[b].nseg;
        active[b] = (uint32_t) lane < nseg[b];
        l[b] = active[b] ? (uint32_t) lane : 0u;
        if (ps[b].flags & SPX_PASSF_INLINE) {
            // (the pass' only descriptor came with its header; its slot entry is needed last)
            q[b].x = (uint32_t) ps[b].mask;
            q[b].y = (uint32_t) (ps[b].mask >> 32);
            slot0[b] = a.descs[rb.desc_off + (uint32_t) ps[b].rank0 + 1u].col0;
        } else
        {
            const uint64_t mk = (ps[b].flags & SPX_PASSF_INLINE) ? 0ull : ps[b].mask;
            const uint32_t rank = (uint32_t) ps[b].rank0 + 2u * (active[b] ? starts_upto(mk, lane) : 0u);
            q[b] = ld_stream(reinterpret_cast<const uint2 *>(a.descs + rb.desc_off + rank));
            slot0[b] = a.descs[rb.desc_off + rank + 1u].col0;
        }
    }
    double v[B][W];
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const double *vals = a.values + rb.val_off + ps[b].val_off;
#pragma unroll
        for (int p = 0; p < W / 2; ++p) {
            const double2 vv = ld_stream(reinterpret_cast<const double2 *>(vals + (uint32_t) p * 2u * nseg[b] + l[b] * 2u));
            v[b][2 * p] = vv.x;
            v[b][2 * p + 1] = vv.y;
        }
        if (W & 1) v[b][W - 1] = ld_stream(vals + (uint32_t) (W / 2) * 2u * nseg[b] + l[b]);
    }
    int row[B], sdc[B];
    uint32_t col[B];
    double xr[B], x[B][W];
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const uint32_t bits = q[b].y;
        const int s = (int) ((ps[b].seg0 + l[b] - ((bits >> 9) & 8191u)) & 0xffffu);
        const uint32_t kind = (bits >> 22) & 7u;
        const int step = (int) (bits >> 25);
        const int drow = kind == SPX_KIND_BLOCK ? 1 : (kind >= SPX_KIND_VERT ? step : 0);
        const int dcol = (kind == SPX_KIND_HORIZ || kind == SPX_KIND_DIAG) ? step : (kind == SPX_KIND_ADIAG ? -step : 0);
        row[b] = (int) (ps[b].elem0 + (bits & 511u)) + s * drow;
        sdc[b] = s * dcol;
        col[b] = q[b].x + (uint32_t) sdc[b];
        const double *xp = a.x + col[b];
        // (x in unaligned pairs, as the unit passes load it, measured 2.3 % slower here: one load per column)
        if (abl::sym_no_x) {
            xr[b] = a.x[lane];
#pragma unroll
            for (int w = 0; w < W; ++w) x[b][w] = a.x[lane + w];
        } else {
            xr[b] = a.x[rb.row0 + (uint32_t) row[b]];
#pragma unroll
            for (int w = 0; w < W; ++w) x[b][w] = xp[w];
        }
    }
#pragma unroll
    for (int b = 0; b < B; ++b) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < W; ++w) t = fma(v[b][w], x[b][w], t);
        if (!active[b]) continue;
        atomicAdd(&tile[row[b]], t);
        if (slot0[b] != SPX_NO_SLOT) {
            double *sl = slots + slot0[b] + (uint32_t) sdc[b];
            if (abl::sym_no_slot_add || abl::sym_one_add) {
                double u = 0.0;
#pragma unroll
                for (int w = 0; w < W; ++w) u += v[b][w] * xr[b];
                if (abl::sym_one_add) atomicAdd(&sl[0], u);
                else if (u == 1.2345e-300) sl[0] = u;
            } else {
#pragma unroll
                for (int w = 0; w < W; ++w) atomicAdd(&sl[w], v[b][w] * xr[b]);
            }
        } else {
            double *yp = a.y + col[b];
#pragma unroll
            for (int w = 0; w < W; ++w) atomicAdd(&yp[w], a.alpha * (v[b][w] * xr[b]));
        }
    }
}

__device__ __forceinline__ void run_symseg(const KernelArgs &a, const SpxRowBlock &rb, const SpxPass &ps,
                                           double *slots, double *tile, int lane)
{
    switch (ps.width) {            // wave-uniform
    case 2: symseg_passes<2, 1>(a, rb, {ps}, slots, tile, lane); break;
    case 3: symseg_passes<3, 1>(a, rb, {ps}, slots, tile, lane); break;
    case 4: symseg_passes<4, 1>(a, rb, {ps}, slots, tile, lane); break;
    case 5: symseg_passes<5, 1>(a, rb, {ps}, slots, tile, lane); break;
    case 6: symseg_passes<6, 1>(a, rb, {ps}, slots, tile, lane); break;
    case 7: symseg_passes<7, 1>(a, rb, {ps}, slots, tile, lane); break;
    default: symseg_passes<8, 1>(a, rb, {ps}, slots, tile, lane); break;
    }
}

// two read-once passes of the same width (<= 4: the registers of two wider ones would cost
// the kernel its eight wavefronts per SIMD) side by side
__device__ __forceinline__ bool run_symseg2(const KernelArgs &a, const SpxRowBlock &rb, const SpxPass &p0,
                                            const SpxPass &p1, double *slots, double *tile, int lane)
{
    if (p0.width != p1.width || p0.width > 4) return false;
    switch (p0.width) {
    case 2: symseg_passes<2, 2>(a, rb, {p0, p1}, slots, tile, lane); break;
    case 3: symseg_passes<3, 2>(a, rb, {p0, p1}, slots, tile, lane); break;
    default: symseg_passes<4, 2>(a, rb, {p0, p1}, slots, tile, lane); break;
    }
    return true;
}

}  // namespace spx
