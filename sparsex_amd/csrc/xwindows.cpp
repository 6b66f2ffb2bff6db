// xwindows.cpp -- see xwindows.hpp.
#include "xwindows.hpp"

#include "threads.hpp"

#include <algorithm>
#include <atomic>
#include <cstring>

namespace spx {

namespace {

struct Interval { int64_t lo, hi; };

// column step per segment of a unit descriptor (gpu_format.h)
inline int desc_dcol(uint32_t bits)
{
    const uint32_t kind = (bits >> 22) & 7u;
    const int step = (int) (bits >> 25);
    return (kind == SPX_KIND_HORIZ || kind == SPX_KIND_DIAG) ? step : (kind == SPX_KIND_ADIAG ? -step : 0);
}

// calls fn(rank, first lane, last lane) for every unit of a unit pass (ranks relative to the row-block)
template <typename F>
inline void for_units_of_pass(const SpxPass &ps, F fn)
{
    const uint64_t mask = spx_pass_mask(&ps);
    uint32_t rank = ps.rank0, first = 0;
    for (uint32_t l = 1; l < ps.nseg; ++l) {
        if ((mask >> l) & 1ull) {
            fn(rank, first, l - 1);
            ++rank;
            first = l;
        }
    }
    fn(rank, first, (uint32_t) ps.nseg - 1u);
}

struct RbResult {
    bool has_units = false, windows = false;
    uint32_t total = 0;
    uint64_t elems = 0;
};

RbResult plan_rowblock(const GpuStream &s, size_t rb_idx, size_t ncols, uint32_t budget, uint32_t gap, XwPlan &plan)
{
    RbResult res;
    const SpxRowBlock &rb = s.rbs[rb_idx];
    const SpxPass *ps0 = s.passes.data() + rb.pass_off;
    // column interval of every descriptor the unit passes use, from the lanes that are there
    std::vector<Interval> span;
    for (uint32_t t = 0; t < rb.n_pass; ++t) {
        const SpxPass &ps = ps0[t];
        if (ps.kind != SPX_PASS_UNIT || ps.nseg == 0) continue;
        res.has_units = true;
        res.elems += (uint64_t) ps.nseg * ps.width;
        for_units_of_pass(ps, [&](uint32_t rank, uint32_t la, uint32_t lb) {
            const SpxUnitDesc &d = s.descs[(size_t) rb.desc_off + rank];
            const int dcol = desc_dcol(d.bits);
            const uint32_t sstart = (d.bits >> 9) & 8191u;
            const int64_t sa = (int64_t) (((uint32_t) ps.seg0 + la - sstart) & 0xffffu);
            const int64_t sb = sa + (int64_t) (lb - la);
            const int64_t ca = (int64_t) d.col0 + sa * dcol, cb = (int64_t) d.col0 + sb * dcol;
            const Interval iv{std::min(ca, cb), std::max(ca, cb) + (int64_t) ps.width};
            if (span.size() <= rank) span.resize((size_t) rank + 1, Interval{INT64_MAX, INT64_MIN});
            span[rank].lo = std::min(span[rank].lo, iv.lo);
            span[rank].hi = std::max(span[rank].hi, iv.hi);
        });
    }
    if (!res.has_units || budget == 0) return res;
    // windows start and end on even columns (16-byte loads and LDS stores), the last one may end
    // with the vector
    std::vector<Interval> iv;
    iv.reserve(span.size());
    for (const Interval &v : span) {
        if (v.lo > v.hi) continue;                            // (a descriptor no unit pass uses)
        if (v.lo < 0 || v.hi > (int64_t) ncols) return res;   // (never in a valid stream: leave the row-block alone)
        iv.push_back(Interval{v.lo & ~(int64_t) 1, std::min<int64_t>((v.hi + 1) & ~(int64_t) 1, (int64_t) ncols)});
    }
    if (iv.empty()) return res;
    std::sort(iv.begin(), iv.end(), [](const Interval &a, const Interval &b) { return a.lo < b.lo; });
    std::vector<Interval> win;
    for (const Interval &v : iv) {
        if (!win.empty() && v.lo <= win.back().hi + (int64_t) gap) win.back().hi = std::max(win.back().hi, v.hi);
        else win.push_back(v);
    }
    // too many pieces: close the smallest gaps first
    while (win.size() > XW_MAX) {
        size_t best = 1;
        for (size_t k = 2; k < win.size(); ++k)
            if (win[k].lo - win[k - 1].hi < win[best].lo - win[best - 1].hi) best = k;
        win[best - 1].hi = win[best].hi;
        win.erase(win.begin() + (std::ptrdiff_t) best);
    }
    uint64_t total = 0;
    for (const Interval &w : win) total += (uint64_t) ((w.hi - w.lo + 1) & ~(int64_t) 1);
    if (total > budget || total > 65534u) return res;
    // the LDS offset of every window, and -- before anything of the plan is written -- of every descriptor's
    // col0 (the column of its segment 0): a descriptor whose col0 lies in no window (its segment 0 is not
    // among this row-block's unit passes; never in a stream of this emitter) leaves the row-block on the
    // plain path as a whole, not half planned
    std::vector<uint32_t> off(win.size());
    uint32_t at = 0;
    for (size_t k = 0; k < win.size(); ++k) {
        off[k] = at;
        at += ((uint32_t) (win[k].hi - win[k].lo) + 1u) & ~1u;
    }
    std::vector<uint32_t> xcol(span.size(), 0u);
    for (size_t rank = 0; rank < span.size(); ++rank) {
        if (span[rank].lo > span[rank].hi) continue;
        const int64_t c0 = (int64_t) s.descs[(size_t) rb.desc_off + rank].col0;
        const size_t above = (size_t) (std::upper_bound(win.begin(), win.end(), c0,
                                                        [](int64_t c, const Interval &w) { return c < w.lo; }) - win.begin());
        if (above == 0 || c0 >= win[above - 1].hi) return res;
        xcol[rank] = (uint32_t) (c0 - win[above - 1].lo + (int64_t) off[above - 1]);
    }
    // accepted: the table, then every descriptor and unit pass of the row-block
    XwEntry *tab = plan.tab.data() + rb_idx * XW_TAB + XW_RANGES;
    for (size_t k = 0; k < win.size(); ++k) {
        tab[k].base = (uint32_t) win[k].lo;
        tab[k].off_len = off[k] | ((uint32_t) (win[k].hi - win[k].lo) << 16);
    }
    for (size_t rank = 0; rank < span.size(); ++rank)
        if (span[rank].lo <= span[rank].hi) plan.xdescs[(size_t) rb.desc_off + rank].col0 = xcol[rank];
    SpxPass *px = plan.passes.data() + rb.pass_off;
    for (uint32_t t = 0; t < rb.n_pass; ++t) {
        SpxPass &ps = px[t];
        if (ps.kind != SPX_PASS_UNIT || ps.nseg == 0) continue;
        ps.flags |= SPX_PASSF_XLDS;
        if (ps.flags & SPX_PASSF_INLINE) {
            const SpxUnitDesc &d = plan.xdescs[(size_t) rb.desc_off + ps.rank0];
            ps.mask = (uint64_t) d.col0 | ((uint64_t) d.bits << 32);
        }
    }
    // the pass range of the pipeline: the longest run of unit passes of width <= 4 that read LDS
    {
        uint32_t best_lo = 0, best_hi = 0, t = 0;
        while (t < rb.n_pass) {
            uint32_t e = t;
            while (e < rb.n_pass && px[e].kind == SPX_PASS_UNIT && (px[e].flags & SPX_PASSF_XLDS) && px[e].width <= 4 &&
                   px[e].nseg > 0)
                ++e;
            if (e - t > best_hi - best_lo) { best_lo = t; best_hi = e; }
            t = e > t ? e : t + 1;
        }
        XwEntry *rg = plan.tab.data() + rb_idx * XW_TAB;
        rg[0].base = best_lo | (best_hi << 16);
        rg[0].off_len = at;
    }
    res.windows = true;
    res.total = at;
    return res;
}

}  // namespace

void plan_unit_xwindows(const GpuStream &s, size_t ncols, uint32_t budget, uint32_t gap, XwPlan &plan,
                        unsigned nthreads)
{
    const size_t n = s.rbs.size();
    plan.tab.assign(n * XW_TAB, XwEntry{0u, 0u});
    plan.xdescs = s.descs;
    plan.passes = s.passes;
    plan.lds_doubles = SPX_MAX_RB_ROWS;
    plan.n_rb_windows = plan.n_rb_units = 0;
    plan.staged_doubles = plan.unit_elems = plan.unit_elems_lds = 0;
    constexpr size_t CHUNK = 128;
    const size_t n_chunks = (n + CHUNK - 1) / CHUNK;
    std::vector<uint32_t> lds_of(n_chunks, 0);
    std::vector<uint64_t> staged_of(n_chunks, 0), elems_of(n_chunks, 0), elems_lds_of(n_chunks, 0);
    std::vector<size_t> nwin_of(n_chunks, 0), nunit_of(n_chunks, 0);
    parallel_for(n_chunks, nthreads, [&](size_t c) {
        const size_t lo = c * CHUNK, hi = std::min(n, lo + CHUNK);
        for (size_t i = lo; i < hi; ++i) {
            const RbResult r = plan_rowblock(s, i, ncols, budget, gap, plan);
            const SpxRowBlock &rb = s.rbs[i];
            // y tile, leftover window, then (on an even offset) the unit windows
            const uint32_t front = ((uint32_t) rb.n_rows + rb.xwin_len + 1u) & ~1u;
            lds_of[c] = std::max(lds_of[c], front + r.total + 3u * (s.pass_stride + 4u * 8u) + 8u);
            staged_of[c] += r.total;
            elems_of[c] += r.elems;
            if (r.windows) elems_lds_of[c] += r.elems;
            nwin_of[c] += r.windows ? 1u : 0u;
            nunit_of[c] += r.has_units ? 1u : 0u;
        }
    });
    for (size_t c = 0; c < n_chunks; ++c) {
        plan.lds_doubles = std::max(plan.lds_doubles, lds_of[c]);
        plan.staged_doubles += staged_of[c];
        plan.unit_elems += elems_of[c];
        plan.unit_elems_lds += elems_lds_of[c];
        plan.n_rb_windows += nwin_of[c];
        plan.n_rb_units += nunit_of[c];
    }
}

}  // namespace spx
