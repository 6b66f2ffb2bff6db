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

// leftover passes whose columns went into the windows: {pass index, first entry in `desc`} and, per half of
// the pass' width and lane, {LDS offsets of its two columns, row | valid columns << 16}
struct GatherOut {
    std::vector<std::pair<uint32_t, uint32_t>> pass_base;
    std::vector<uint32_t> desc;         // two words per entry
};

// column of leftover e of a gather pass that gathers through L2 (SPX_PASS_GATHER)
inline int64_t leftover_col(const GpuStream &s, const SpxRowBlock &rb, size_t e)
{
    const uint8_t *c = s.cidx.data() + (size_t) rb.cidx_off * 16u;
    if (rb.cidx_width == 4) {
        uint32_t v;
        std::memcpy(&v, c + e * 4, 4);
        return (int64_t) rb.cbase + v;
    }
    uint16_t v;
    std::memcpy(&v, c + e * 2, 2);
    if (rb.cidx_width == 3) return (int64_t) rb.cbase + (v | ((uint32_t) c[(size_t) rb.hi_off * 16u + e] << 16));
    return (int64_t) rb.cbase + v;
}

RbResult plan_rowblock(const GpuStream &s, size_t rb_idx, size_t ncols, uint32_t budget, uint32_t gap, bool with_leftovers,
                       XwPlan &plan, GatherOut &gout)
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
    // ... and the columns of the leftovers (SPX_PASS_GATHER), where they fit as well: their passes then read x
    // from LDS too (persistent kernel: as rounds of the pipeline)
    std::vector<int64_t> left;
    for (uint32_t t = 0; t < rb.n_pass && with_leftovers; ++t) {
        const SpxPass &ps = ps0[t];
        if (ps.kind != SPX_PASS_GATHER) continue;
        for (uint32_t l = 0; l < ps.nseg; ++l) {
            const uint32_t len = SPX_SEGROW_LEN(s.segrows[(size_t) rb.seg_off + ps.seg0 + l]);
            for (uint32_t w = 0; w < len && w < ps.width; ++w) left.push_back(leftover_col(s, rb, (size_t) ps.elem0 + l + (size_t) w * ps.nseg));
        }
    }
    auto build = [&](bool with_left, std::vector<Interval> &win) -> uint64_t {
        std::vector<Interval> all = iv;
        if (with_left)
            for (int64_t c : left) {
                if (c < 0 || c >= (int64_t) ncols) return UINT64_MAX;
                all.push_back(Interval{c & ~(int64_t) 1, std::min<int64_t>((c + 2) & ~(int64_t) 1, (int64_t) ncols)});
            }
        std::sort(all.begin(), all.end(), [](const Interval &a, const Interval &b) { return a.lo < b.lo; });
        win.clear();
        for (const Interval &v : all) {
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
        return total;
    };
    std::vector<Interval> win;
    bool left_in = !left.empty();
    uint64_t total = left_in ? build(true, win) : UINT64_MAX;
    if (total > budget || total > 65534u) {
        left_in = false;
        total = build(false, win);
    }
    if (total > budget || total > 65534u) return res;
    // accepted: the table, then every descriptor and unit pass of the row-block
    XwEntry *tab = plan.tab.data() + rb_idx * XW_TAB + XW_RANGES;
    std::vector<uint32_t> off(win.size());
    uint32_t at = 0;
    for (size_t k = 0; k < win.size(); ++k) {
        const uint32_t len = (uint32_t) (win[k].hi - win[k].lo);
        off[k] = at;
        tab[k].base = (uint32_t) win[k].lo;
        tab[k].off_len = at | (len << 16);
        at += (len + 1u) & ~1u;
    }
    for (size_t rank = 0; rank < span.size(); ++rank) {
        if (span[rank].lo > span[rank].hi) continue;
        SpxUnitDesc &d = plan.xdescs[(size_t) rb.desc_off + rank];
        size_t k = (size_t) (std::upper_bound(win.begin(), win.end(), (int64_t) d.col0,
                                              [](int64_t c, const Interval &w) { return c < w.lo; }) - win.begin()) - 1u;
        d.col0 = (uint32_t) ((int64_t) d.col0 - win[k].lo + (int64_t) off[k]);
    }
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
    // leftover passes whose columns are in the windows: per half of the width and lane {two LDS offsets, row | valid << 16}
    if (left_in) {
        auto lds_of = [&](int64_t c) -> uint32_t {
            const size_t k = (size_t) (std::upper_bound(win.begin(), win.end(), c, [](int64_t cc, const Interval &w) { return cc < w.lo; }) - win.begin()) - 1u;
            return (uint32_t) (c - win[k].lo + (int64_t) off[k]);
        };
        for (uint32_t t = 0; t < rb.n_pass; ++t) {
            const SpxPass &ps = ps0[t];
            if (ps.kind != SPX_PASS_GATHER || ps.nseg == 0) continue;
            gout.pass_base.emplace_back((uint32_t) (rb.pass_off + t), (uint32_t) (gout.desc.size() / 2));
            px[t].flags |= SPX_PASSF_XLDS;          // (the other kernels do not look at the flags of a leftover pass)
            const uint32_t halves = ((uint32_t) ps.width + 1u) / 2u;
            for (uint32_t h = 0; h < halves; ++h)
                for (uint32_t l = 0; l < ps.nseg; ++l) {
                    const uint32_t sr = s.segrows[(size_t) rb.seg_off + ps.seg0 + l];
                    const uint32_t len = SPX_SEGROW_LEN(sr), row = SPX_SEGROW_ROW(sr);
                    uint32_t o[2] = {0u, 0u}, valid = 0;
                    for (uint32_t j = 0; j < 2; ++j) {
                        const uint32_t w = 2u * h + j;
                        if (w < len && w < ps.width) {
                            o[j] = lds_of(leftover_col(s, rb, (size_t) ps.elem0 + l + (size_t) w * ps.nseg));
                            valid = j + 1u;
                        }
                    }
                    gout.desc.push_back(o[0] | (o[1] << 16));
                    gout.desc.push_back(row | (valid << 16));
                }
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
                        unsigned nthreads, bool with_leftovers)
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
    std::vector<GatherOut> gout_of(n_chunks);
    parallel_for(n_chunks, nthreads, [&](size_t c) {
        const size_t lo = c * CHUNK, hi = std::min(n, lo + CHUNK);
        for (size_t i = lo; i < hi; ++i) {
            const RbResult r = plan_rowblock(s, i, ncols, budget, gap, with_leftovers, plan, gout_of[c]);
            const SpxRowBlock &rb = s.rbs[i];
            // (the spare entry: what the persistent kernel's write-out needs of the row-block header)
            plan.tab[i * XW_TAB + 1] = XwEntry{rb.row0, rb.n_rows};
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
    plan.gather_base.assign(s.passes.size(), UINT32_MAX);
    plan.gdesc.clear();
    for (size_t c = 0; c < n_chunks; ++c) {
        const uint32_t at = (uint32_t) (plan.gdesc.size() / 2);
        for (const auto &pb : gout_of[c].pass_base) plan.gather_base[pb.first] = at + pb.second;
        plan.gdesc.insert(plan.gdesc.end(), gout_of[c].desc.begin(), gout_of[c].desc.end());
    }
    for (size_t c = 0; c < n_chunks; ++c) {
        plan.lds_doubles = std::max(plan.lds_doubles, lds_of[c]);
        plan.staged_doubles += staged_of[c];
        plan.unit_elems += elems_of[c];
        plan.unit_elems_lds += elems_lds_of[c];
        plan.n_rb_windows += nwin_of[c];
        plan.n_rb_units += nunit_of[c];
    }
}


namespace {

inline void xwp_empty_pass(XwpRound &r, int p)
{
    for (int k = 0; k < 6; ++k) r.w[6 * p + k] = 0;
    r.w[6 * p + 5] = 1u << 24;          // no lanes, width 1, the first value and descriptor of the stream
    r.w[12 + p] = 0;
}

inline void xwp_set_pass(XwpRound &r, int p, const SpxRowBlock &rb, const SpxPass &ps)
{
    const uint64_t v = rb.val_off + ps.val_off;
    const uint64_t mask = spx_pass_mask(&ps);
    r.w[6 * p + 0] = (uint32_t) v;
    r.w[6 * p + 1] = (uint32_t) (v >> 32);
    r.w[6 * p + 2] = rb.desc_off + ps.rank0;
    r.w[6 * p + 3] = (uint32_t) mask;
    r.w[6 * p + 4] = (uint32_t) (mask >> 32);
    r.w[6 * p + 5] = (uint32_t) ps.seg0 | ((uint32_t) ps.nseg << 16) | ((uint32_t) ps.width << 24);
    r.w[12 + p] = ps.elem0;
}

}  // namespace

void plan_persistent_rounds(const GpuStream &s, const XwPlan &plan, const uint32_t first[9], uint32_t waves,
                            uint32_t wgs_per_xcd, uint32_t tail_rounds, XwpPlan &out, unsigned nthreads)
{
    out = XwpPlan();
    out.waves = waves;
    out.wgs_per_xcd = wgs_per_xcd;
    out.tail_rounds = tail_rounds;
    if (s.rbs.empty() || !s.shared.empty() || waves == 0 || wgs_per_xcd == 0) return;
    for (const SpxRowBlock &rb : s.rbs) {
        if (rb.xwin_len || (rb.flags & (SPX_RB_SHARED | SPX_RB_ACCUM | SPX_RB_PHASE_START))) return;
        out.max_rows = std::max<uint32_t>(out.max_rows, rb.n_rows);
    }
    for (size_t i = 0; i < s.rbs.size(); ++i) {
        out.max_window = std::max(out.max_window, plan.tab[i * XW_TAB].off_len);
        // (a window of odd length ends the vector; its last 16-byte piece would read past the end of x)
        for (uint32_t k = 0; k < XW_MAX; ++k)
            if ((plan.tab[i * XW_TAB + XW_RANGES + k].off_len >> 16) & 1u) return;
    }
    if (out.max_window > 4096u) return;
    const size_t n_wg = 8u * (size_t) wgs_per_xcd, n_streams = n_wg * waves;
    // pass 1: rounds per list
    out.stream_len.assign(n_streams, 0);
    auto rounds_of = [&](size_t rb_idx, uint32_t w) -> uint32_t {
        const SpxRowBlock &rb = s.rbs[rb_idx];
        const uint32_t range = plan.tab[rb_idx * XW_TAB].base, lo = range & 0xffffu, hi = range >> 16;
        const SpxPass *ps = plan.passes.data() + rb.pass_off;
        uint32_t n_in = 0;
        for (uint32_t t = w; t < rb.n_pass; t += waves) {
            if (t >= lo && t < hi) n_in += 1u;
            else if (ps[t].kind == SPX_PASS_GATHER && plan.gather_base[rb.pass_off + t] != UINT32_MAX) n_in += ((uint32_t) ps[t].width + 1u) / 2u;
        }
        return std::max<uint32_t>(1u, (n_in + 1u) / 2u);
    };
    parallel_for(n_wg, nthreads, [&](size_t b) {
        const uint32_t xcd = (uint32_t) (b & 7u), g = (uint32_t) (b >> 3);
        for (uint32_t w = 0; w < waves; ++w) {
            uint32_t n = 0;
            for (size_t i = (size_t) first[xcd] + g; i < first[xcd + 1]; i += wgs_per_xcd) n += rounds_of(i, w);
            out.stream_len[b * waves + w] = n;
        }
    });
    out.stream_off.assign(n_streams + 1, 0);
    for (size_t k = 0; k < n_streams; ++k) out.stream_off[k + 1] = out.stream_off[k] + out.stream_len[k] + tail_rounds;
    out.rounds.resize(out.stream_off[n_streams]);
    // pass 2: the lists
    std::atomic<uint64_t> n_generic_all(0);
    parallel_for(n_wg, nthreads, [&](size_t b) {
        const uint32_t xcd = (uint32_t) (b & 7u), g = (uint32_t) (b >> 3);
        uint64_t n_generic = 0;
        for (uint32_t w = 0; w < waves; ++w) {
            XwpRound *at = out.rounds.data() + out.stream_off[b * waves + w];
            for (size_t i = (size_t) first[xcd] + g; i < first[xcd + 1]; i += wgs_per_xcd) {
                const SpxRowBlock &rb = s.rbs[i];
                const SpxPass *ps = plan.passes.data() + rb.pass_off;
                const uint32_t range = plan.tab[i * XW_TAB].base, lo = range & 0xffffu, hi = range >> 16;
                bool generic = false;
                XwpRound *first_round = at;
                int half = 0;
                auto put = [&](const XwpRound &one) {       // (one.w[0..5], w[12]: a pass in slot 0)
                    if (half == 0) {
                        xwp_empty_pass(*at, 1);
                        for (int k = 0; k < 6; ++k) at->w[k] = one.w[k];
                        at->w[12] = one.w[12];
                        // (the empty half of a round reads where the first half does: lines that are on their way anyway)
                        for (int k = 0; k < 3; ++k) at->w[6 + k] = at->w[k];
                        if ((one.w[5] >> 24) >= XWP_WIDTH_GATHER2) at->w[8] = 0;     // (... a descriptor of the unit array)
                        at->w[14] = 0;
                        at->w[15] = (uint32_t) i;
                        half = 1;
                    } else {
                        for (int k = 0; k < 6; ++k) at->w[6 + k] = one.w[k];
                        at->w[13] = one.w[12];
                        ++at;
                        half = 0;
                    }
                };
                for (uint32_t t = w; t < rb.n_pass; t += waves) {
                    XwpRound one;
                    if (t >= lo && t < hi) {
                        xwp_set_pass(one, 0, rb, ps[t]);
                        put(one);
                    } else if (ps[t].kind == SPX_PASS_GATHER && plan.gather_base[rb.pass_off + t] != UINT32_MAX) {
                        // a leftover pass whose columns are in the windows: one pass of the pipeline per two
                        // nonzeros of its lanes -- their values lie pair by pair as a unit pass' do
                        const uint32_t halves = ((uint32_t) ps[t].width + 1u) / 2u, nseg = ps[t].nseg;
                        for (uint32_t h = 0; h < halves; ++h) {
                            const bool single = (ps[t].width & 1u) && h + 1u == halves;
                            const uint64_t v = rb.val_off + ps[t].val_off + (uint64_t) h * 2u * nseg;
                            one.w[0] = (uint32_t) v;
                            one.w[1] = (uint32_t) (v >> 32);
                            one.w[2] = (uint32_t) plan.xdescs.size() + plan.gather_base[rb.pass_off + t] + h * nseg;
                            one.w[3] = one.w[4] = 0;
                            one.w[5] = (nseg << 16) | ((single ? XWP_WIDTH_GATHER1 : XWP_WIDTH_GATHER2) << 24);
                            one.w[12] = 0;
                            put(one);
                        }
                    } else {
                        generic = true;
                        ++n_generic;
                    }
                }
                if (half) ++at;
                if (at == first_round) {             // nothing of this wavefront's runs through the pipeline here
                    xwp_empty_pass(*at, 0);
                    xwp_empty_pass(*at, 1);
                    at->w[14] = 0;
                    at->w[15] = (uint32_t) i;
                    ++at;
                }
                at[-1].w[14] |= XWP_LAST | (generic ? XWP_GENERIC : 0u);
            }
            for (uint32_t k = 0; k < tail_rounds; ++k, ++at) {
                xwp_empty_pass(*at, 0);
                xwp_empty_pass(*at, 1);
                at->w[14] = 0;
                at->w[15] = 0;
            }
            if ((size_t) (at - out.rounds.data()) != out.stream_off[b * waves + w + 1]) throw FatalError("persistent rounds: list length");
        }
        n_generic_all += n_generic;
    });
    // passes that stay outside the pipeline stall their whole workgroup at the end of their row-block: a few
    // are tolerable, a stream of them (leftovers all over x) is what the other kernels are for
    out.generic_passes = n_generic_all.load();
    size_t n_pass_all = 0;
    for (const SpxRowBlock &rb : s.rbs) n_pass_all += rb.n_pass;
    out.usable = out.generic_passes * 50u <= n_pass_all;
}

}  // namespace spx
