// stream_index.cpp -- see stream_index.hpp.
#include "stream_index.hpp"

#include "threads.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace spx {

namespace {

inline uint32_t popcount_upto(uint64_t mask, uint32_t lane)
{
    // set bits in lanes 1..lane (bit 0 is never set)
    const uint64_t m = lane >= 63 ? mask : (mask & ((2ull << lane) - 1ull));
    return (uint32_t) __builtin_popcountll(m);
}

struct LaneSeg { int64_t row; int64_t col0; int64_t dcol_elem; bool ok; };

// row (relative to the row-block) and first column of lane l of a unit pass
inline void unit_lane(const GpuStream &s, const SpxRowBlock &rb, const SpxPass &ps, uint32_t l,
                      int64_t &row, int64_t &col, uint32_t *slot = nullptr)
{
    // (read-once symmetric segments: two entries per unit, the second holds the slot)
    const uint32_t stride = ps.kind == SPX_PASS_SYMSEG ? 2u : 1u;
    const uint32_t rank = (uint32_t) ps.rank0 + stride * popcount_upto(spx_pass_mask(&ps), l);
    const SpxUnitDesc &d = s.descs[(size_t) rb.desc_off + rank];
    if (slot) *slot = stride == 2 ? s.descs[(size_t) rb.desc_off + rank + 1].col0 : SPX_NO_SLOT;
    const uint32_t bits = d.bits;
    const int sidx = (int) ((ps.seg0 + l - ((bits >> 9) & 8191u)) & 0xffffu);
    const uint32_t kind = (bits >> 22) & 7u;
    const int step = (int) (bits >> 25);
    const int drow = kind == SPX_KIND_BLOCK ? 1 : (kind >= SPX_KIND_VERT ? step : 0);
    const int dcol = (kind == SPX_KIND_HORIZ || kind == SPX_KIND_DIAG)
                         ? step : (kind == SPX_KIND_ADIAG ? -step : 0);
    row = (int64_t) ps.elem0 + (int64_t) (bits & 511u) + (int64_t) sidx * drow;
    col = (int64_t) d.col0 + (int64_t) sidx * dcol;
    if (slot && *slot != SPX_NO_SLOT) *slot = (uint32_t)((int64_t) *slot + (int64_t) sidx * dcol);
}

// column of leftover e of a gather pass (through L2, or from the x window)
inline int64_t gather_col(const GpuStream &s, const SpxRowBlock &rb, const SpxPass &ps, size_t e)
{
    if (ps.kind == SPX_PASS_GATHER_LDS) {
        uint16_t v;
        std::memcpy(&v, s.cidx.data() + ((size_t) rb.cidx_off + rb.near_off) * 16u + e * 2, 2);
        return (int64_t) rb.xwin_base + v;
    }
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

inline bool is_gather(const SpxPass &ps)
{
    return ps.kind == SPX_PASS_GATHER || ps.kind == SPX_PASS_GATHER_LDS;
}

}  // namespace

void stream_locate(const GpuStream &s, idx_t row, idx_t col, std::vector<size_t> &out)
{
    // row-blocks are emitted in ascending row order -- inside every column phase (most streams
    // hold one): in each, the first row-block that ends behind `row`
    std::vector<size_t> starts(1, 0);
    for (size_t i = 1; i < s.rbs.size(); ++i)
        if (s.rbs[i].flags & SPX_RB_PHASE_START) starts.push_back(i);
    starts.push_back(s.rbs.size());
    for (size_t ph = 0; ph + 1 < starts.size(); ++ph) {
    size_t lo = starts[ph], hi = starts[ph + 1];
    const size_t end = hi;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if ((int64_t) s.rbs[mid].row0 + s.rbs[mid].n_rows <= (int64_t) row) lo = mid + 1;
        else hi = mid;
    }
    for (size_t i = lo; i < end && (int64_t) s.rbs[i].row0 <= (int64_t) row; ++i) {
        const SpxRowBlock &rb = s.rbs[i];
        if ((int64_t) row >= (int64_t) rb.row0 + rb.n_rows) continue;
        const int64_t rrel = (int64_t) row - rb.row0;
        for (uint32_t t = 0; t < rb.n_pass; ++t) {
            const SpxPass &ps = s.passes[(size_t) rb.pass_off + t];
            const uint32_t nseg = ps.nseg, W = ps.width;
            const size_t vbase = (size_t) rb.val_off + ps.val_off;
            if (is_gather(ps)) {
                for (uint32_t l = 0; l < nseg; ++l) {
                    const uint32_t sr = s.segrows[(size_t) rb.seg_off + ps.seg0 + l];
                    if ((int64_t) SPX_SEGROW_ROW(sr) != rrel) continue;
                    for (uint32_t w = 0; w < W && w < SPX_SEGROW_LEN(sr); ++w) {
                        const int64_t c = gather_col(s, rb, ps, (size_t) ps.elem0 + l + (size_t) w * nseg);
                        if (c == (int64_t) col) out.push_back(vbase + spx_pass_value_index(l, w, nseg, W));
                    }
                }
            } else if (ps.kind == SPX_PASS_SYMTILE) {
                for (uint32_t l = 0; l < nseg; ++l) {
                    const SpxUnitDesc &d = s.descs[(size_t) rb.desc_off + ps.rank0 + (l >> 3)];
                    const int64_t r = (int64_t) ps.elem0 + (int64_t) (d.bits & 511u) + (l & 7u);
                    if (r != rrel) continue;
                    const int64_t w = (int64_t) col - (int64_t) d.col0;
                    if (w >= 0 && w < 8) out.push_back(vbase + spx_pass_value_index(l, (uint32_t) w, nseg, 8));
                }
            } else {
                for (uint32_t l = 0; l < nseg; ++l) {
                    int64_t r, c0;
                    unit_lane(s, rb, ps, l, r, c0);
                    if (r != rrel) continue;
                    const int64_t w = (int64_t) col - c0;
                    if (w >= 0 && w < (int64_t) W)
                        out.push_back(vbase + spx_pass_value_index(l, (uint32_t) w, nseg, W));
                }
            }
        }
    }
    }
}

void stream_touched_rows(const GpuStream &s, idx_t below, std::vector<idx_t> &rows)
{
    rows.clear();
    if (below <= 0) return;
    std::vector<char> mark((size_t) below, 0);
    for (const SpxRowBlock &rb : s.rbs) {
        for (uint32_t t = 0; t < rb.n_pass; ++t) {
            const SpxPass &ps = s.passes[(size_t) rb.pass_off + t];
            // (row-blocks of own rows matter through what they hand to rows in front of them)
            if ((int64_t) rb.row0 >= (int64_t) below && ps.kind != SPX_PASS_SYMSEG && ps.kind != SPX_PASS_SYMTILE) continue;
            for (uint32_t l = 0; l < ps.nseg; ++l) {
                int64_t r;
                if (is_gather(ps)) {
                    r = SPX_SEGROW_ROW(s.segrows[(size_t) rb.seg_off + ps.seg0 + l]);
                } else if (ps.kind == SPX_PASS_SYMTILE) {
                    const SpxUnitDesc &d = s.descs[(size_t) rb.desc_off + ps.rank0 + (l >> 3)];
                    r = (int64_t) ps.elem0 + (int64_t) (d.bits & 511u) + (l & 7u);
                    const int64_t c = (int64_t) d.col0 + (l & 7u);   // the transposed tile's row
                    if (c < (int64_t) below) mark[(size_t) c] = 1;
                } else {
                    int64_t c;
                    unit_lane(s, rb, ps, l, r, c);
                    if (ps.kind == SPX_PASS_SYMSEG)                  // through a slot or straight into y
                        for (uint32_t w = 0; w < ps.width; ++w)
                            if (c + w < (int64_t) below) mark[(size_t)(c + w)] = 1;
                }
                r += rb.row0;
                if (r < (int64_t) below) mark[(size_t) r] = 1;
            }
        }
    }
    for (uint32_t r : s.mirror_rows)
        if ((int64_t) r < (int64_t) below) mark[r] = 1;
    // (a restored stream: the per-row lists of the spilled sums are what is left)
    for (size_t r = 0; r + 1 < s.fix_ptr.size() && r < (size_t) below; ++r)
        if (s.fix_ptr[r + 1] > s.fix_ptr[r]) mark[r] = 1;
    for (idx_t r = 0; r < below; ++r)
        if (mark[(size_t) r]) rows.push_back(r);
}

void stream_read_cols(const GpuStream &s, idx_t own_lo, idx_t own_hi, size_t ncols, std::vector<idx_t> &cols)
{
    cols.clear();
    std::vector<char> mark(ncols, 0);
    auto hit = [&](int64_t c) {
        if (c >= 0 && (size_t) c < ncols) mark[(size_t) c] = 1;
    };
    for (const SpxRowBlock &rb : s.rbs) {
        for (uint32_t t = 0; t < rb.n_pass; ++t) {
            const SpxPass &ps = s.passes[(size_t) rb.pass_off + t];
            const uint32_t nseg = ps.nseg, W = ps.width;
            for (uint32_t l = 0; l < nseg; ++l) {
                if (is_gather(ps)) {
                    const uint32_t sr = s.segrows[(size_t) rb.seg_off + ps.seg0 + l];
                    for (uint32_t w = 0; w < W && w < SPX_SEGROW_LEN(sr); ++w)
                        hit(gather_col(s, rb, ps, (size_t) ps.elem0 + l + (size_t) w * nseg));
                } else if (ps.kind == SPX_PASS_SYMTILE) {
                    const SpxUnitDesc &d = s.descs[(size_t) rb.desc_off + ps.rank0 + (l >> 3)];
                    for (uint32_t w = 0; w < 8; ++w) hit((int64_t) d.col0 + w);
                    // (the transposed products multiply x of the lane's own row)
                    hit((int64_t) rb.row0 + (int64_t) ps.elem0 + (int64_t) (d.bits & 511u) + (l & 7u));
                } else {
                    int64_t r, c0;
                    unit_lane(s, rb, ps, l, r, c0);
                    for (uint32_t w = 0; w < W; ++w) hit(c0 + w);
                    if (ps.kind == SPX_PASS_SYMSEG) hit((int64_t) rb.row0 + r);
                }
            }
        }
    }
    for (uint32_t c : s.mirror_col) hit((int64_t) c);
    for (size_t c = 0; c < ncols; ++c)
        if (mark[c] && ((idx_t) c < own_lo || (idx_t) c >= own_hi)) cols.push_back((idx_t) c);
}

void stream_rowblock_xpieces(const GpuStream &s, size_t ncols, size_t piece, std::vector<uint64_t> &mask, unsigned nthreads)
{
    const size_t n = s.rbs.size();
    mask.assign(n, 0ull);
    if (!piece || piece * 64 < ncols) throw FatalError("pieces of x: more than 64");
    const int64_t last = ncols ? (int64_t) ncols - 1 : 0;
    constexpr size_t CHUNK = 64;
    parallel_for((n + CHUNK - 1) / CHUNK, nthreads, [&](size_t c) {
        for (size_t i = c * CHUNK; i < std::min(n, (c + 1) * CHUNK); ++i) {
            const SpxRowBlock &rb = s.rbs[i];
            uint64_t m = 0;
            // columns [c0, c1] (clipped to the vector: idle lanes shadow others, padding may point anywhere)
            auto hit = [&](int64_t c0, int64_t c1) {
                c0 = std::min(std::max<int64_t>(c0, 0), last);
                c1 = std::min(std::max<int64_t>(c1, 0), last);
                for (size_t p = (size_t) c0 / piece; p <= (size_t) c1 / piece; ++p) m |= 1ull << p;
            };
            if (rb.xwin_len) hit((int64_t) rb.xwin_base, (int64_t) rb.xwin_base + rb.xwin_len - 1);
            for (uint32_t t = 0; t < rb.n_pass; ++t) {
                const SpxPass &ps = s.passes[(size_t) rb.pass_off + t];
                const uint32_t nseg = ps.nseg, W = ps.width;
                if (ps.kind == SPX_PASS_GATHER_LDS || nseg == 0) continue;      // (the window, above)
                if (is_gather(ps)) {
                    for (uint32_t l = 0; l < nseg; ++l)
                        for (uint32_t w = 0; w < W; ++w) {
                            const int64_t col = gather_col(s, rb, ps, (size_t) ps.elem0 + l + (size_t) w * nseg);
                            hit(col, col);
                        }
                } else if (ps.kind == SPX_PASS_SYMTILE) {
                    for (uint32_t l = 0; l < nseg; l += 8) {
                        const SpxUnitDesc &d = s.descs[(size_t) rb.desc_off + ps.rank0 + (l >> 3)];
                        hit((int64_t) d.col0, (int64_t) d.col0 + 7);
                    }
                } else {
                    // the lanes of a unit step through its columns evenly: its first and its last lane bound them
                    const uint64_t starts = spx_pass_mask(&ps);
                    uint32_t first = 0;
                    for (uint32_t l = 1; l <= nseg; ++l) {
                        if (l < nseg && !((starts >> l) & 1ull)) continue;
                        int64_t r, ca, cb;
                        unit_lane(s, rb, ps, first, r, ca);
                        unit_lane(s, rb, ps, l - 1, r, cb);
                        hit(std::min(ca, cb), std::max(ca, cb) + (int64_t) W);     // (+ the value-pair loads' second x of odd widths)
                        first = l;
                    }
                }
            }
            // (symmetric streams: the diagonal term and the transposed products read x of the own rows)
            if (rb.n_rows && !s.dvalues.empty()) hit((int64_t) rb.row0, (int64_t) rb.row0 + rb.n_rows - 1);
            mask[i] = m;
        }
    });
    for (uint32_t c : s.mirror_col)
        if (!mask.empty()) mask[0] |= 1ull << ((size_t) std::min<int64_t>((int64_t) c, last) / piece);
}

void stream_locate_mirror(const GpuStream &s, idx_t row, idx_t col, std::vector<size_t> &out)
{
    auto it = std::lower_bound(s.mirror_rows.begin(), s.mirror_rows.end(), (uint32_t) row);
    if (it == s.mirror_rows.end() || *it != (uint32_t) row) return;
    const size_t t = (size_t) (it - s.mirror_rows.begin());
    for (uint32_t k = s.mirror_ptr[t]; k < s.mirror_ptr[t + 1]; ++k)
        if (s.mirror_col[k] == (uint32_t) col) out.push_back(k);
}

namespace {

// first columns of the clusters of unit descriptors of a row-block, relative to its first row
// (clusters: descriptor columns closer together than `gap`)
void band_offsets(const GpuStream &s, const SpxRowBlock &rb, int64_t gap, std::vector<int64_t> &out)
{
    std::vector<int64_t> cols;
    for (uint32_t t = 0; t < rb.n_pass; ++t) {
        const SpxPass &ps = s.passes[(size_t) rb.pass_off + t];
        if (ps.kind != SPX_PASS_UNIT && ps.kind != SPX_PASS_SYMSEG) continue;
        const uint32_t stride = ps.kind == SPX_PASS_SYMSEG ? 2u : 1u;
        const uint32_t n = popcount_upto(spx_pass_mask(&ps), (uint32_t) ps.nseg - 1u) + 1u;
        for (uint32_t k = 0; k < n; ++k)
            cols.push_back((int64_t) s.descs[(size_t) rb.desc_off + ps.rank0 + stride * k].col0);
    }
    out.clear();
    if (cols.empty()) return;
    std::sort(cols.begin(), cols.end());
    out.push_back(cols[0] - (int64_t) rb.row0);
    for (size_t i = 1; i < cols.size(); ++i)
        if (cols[i] - cols[i - 1] > gap) out.push_back(cols[i] - (int64_t) rb.row0);
}

}  // namespace

std::vector<uint32_t> stream_band_order(const GpuStream &s, size_t lo, size_t hi, size_t &stride_rows)
{
    stride_rows = 0;
    std::vector<uint32_t> order;
    if (hi <= lo + 512) return order;                  // (a few hundred row-blocks are in flight at once anyway)
    // 1. the distance: row-blocks with exactly three bands, equally spaced, far apart; most of a
    //    sample must agree on it
    std::vector<int64_t> found, off;
    const size_t samples = 48;
    for (size_t k = 0; k < samples; ++k) {
        const SpxRowBlock &rb = s.rbs[lo + (hi - lo) * (2 * k + 1) / (2 * samples)];
        const int64_t rows = std::max<int64_t>(rb.n_rows, 64);
        band_offsets(s, rb, 4 * rows, off);
        // (three bands in a row at equal distances, far apart; other clusters -- the main
        // diagonal of a KKT system, a few boundary columns -- may sit anywhere around them)
        for (size_t i = 0; i + 2 < off.size() && off.size() <= 8; ++i) {
            const int64_t d1 = off[i + 1] - off[i], d2 = off[i + 2] - off[i + 1];
            if (d1 < 32 * rows || std::llabs(d1 - d2) > 8) continue;
            found.push_back((d1 + d2) / 2);
            break;
        }
    }
    if (found.size() * 4 < samples * 3) return order;
    std::sort(found.begin(), found.end());
    const int64_t S = found[found.size() / 2];
    size_t agree = 0;
    for (int64_t v : found) agree += std::llabs(v - S) <= 8 ? 1 : 0;
    if (agree * 4 < samples * 3 || S <= 0) return order;
    // 2. the order: (strip of the plane, plane, position in the strip)
    const int64_t row_first = (int64_t) s.rbs[lo].row0;
    int64_t rows_total = 0;
    for (size_t i = lo; i < hi; ++i) {
        if ((int64_t) s.rbs[i].row0 < row_first) return std::vector<uint32_t>();     // (not ascending: leave it)
        rows_total += s.rbs[i].n_rows;
    }
    if ((int64_t) s.rbs[hi - 1].row0 - row_first < 3 * S) return order;             // fewer than three planes: nothing to gain
    const int64_t avg_rows = std::max<int64_t>(rows_total / (int64_t)(hi - lo), 1);
    const int64_t strip = std::max<int64_t>(24 * avg_rows, 1);                       // ~24 row-blocks of a plane per strip
    struct Key { int64_t strip, plane, pos; uint32_t idx; };
    std::vector<Key> keys(hi - lo);
    for (size_t i = lo; i < hi; ++i) {
        const int64_t r = (int64_t) s.rbs[i].row0 - row_first;
        keys[i - lo] = Key{(r % S) / strip, r / S, r % S, (uint32_t) i};
    }
    std::sort(keys.begin(), keys.end(), [](const Key &a, const Key &b) {
        if (a.strip != b.strip) return a.strip < b.strip;
        if (a.plane != b.plane) return a.plane < b.plane;
        return a.pos < b.pos;
    });
    order.resize(hi - lo);
    for (size_t k = 0; k < keys.size(); ++k) order[k] = keys[k].idx;
    stride_rows = (size_t) S;
    return order;
}

bool stream_validate(const GpuStream &s, size_t nrows, size_t ncols, size_t n_values,
                     std::string &why)
{
#define SPX_REQUIRE(cond, msg)                                                                   \
    do {                                                                                         \
        if (!(cond)) {                                                                           \
            why = msg;                                                                           \
            return false;                                                                        \
        }                                                                                        \
    } while (0)
    SPX_REQUIRE(s.rbs.empty() || s.pass_stride >= 1, "pass stride missing");
    SPX_REQUIRE(s.passes.size() == s.rbs.size() * (size_t) s.pass_stride, "pass table size");
    SPX_REQUIRE(s.waves == 2 || s.waves == 4 || s.waves == 8, "wavefronts per workgroup");
    // row-block flags are not covered by anything else a file is checked against, and the kernels
    // trust them: SPX_RB_ACCUM on every row-block or on none, with 2, 4 or 8 column slices; a
    // SPX_RB_PRIVATE row-block (it STORES its rows) must be one nobody else adds to -- no slot group
    // of any row-block, no slot-less read-once segment, no mirror list reaches its rows -- and must
    // not be part of an over-long row
    {
        size_t n_accum = 0, n_phase = 0;
        for (size_t i = 0; i < s.rbs.size(); ++i) {
            n_accum += (s.rbs[i].flags & SPX_RB_ACCUM) ? 1 : 0;
            n_phase += (i && (s.rbs[i].flags & SPX_RB_PHASE_START)) ? 1 : 0;
        }
        SPX_REQUIRE(n_accum == 0 || n_accum == s.rbs.size(), "column-slice flag on some row-blocks only");
        SPX_REQUIRE(n_accum == 0 || n_phase == 1 || n_phase == 3 || n_phase == 7, "column slices in one launch: 2, 4 or 8");
        bool any_private = false;
        for (const SpxRowBlock &rb : s.rbs) any_private = any_private || (rb.flags & SPX_RB_PRIVATE);
        if (any_private) {
            std::vector<char> reached(nrows, 0);
            for (uint32_t c : s.slot_group_col)
                for (size_t k = 0; k < 8; ++k)
                    if ((size_t) c + k < nrows) reached[(size_t) c + k] = 1;
            for (uint32_t r : s.mirror_rows)
                if (r < nrows) reached[r] = 1;
            for (const SpxRowBlock &rb : s.rbs) {
                SPX_REQUIRE((size_t) rb.pass_off + rb.n_pass <= s.passes.size(), "pass range");
                for (uint32_t t = 0; t < rb.n_pass; ++t) {
                    const SpxPass &ps = s.passes[(size_t) rb.pass_off + t];
                    if (ps.kind != SPX_PASS_SYMSEG) continue;
                    for (uint32_t l = 0; l < ps.nseg; ++l) {
                        const uint32_t rank = (uint32_t) ps.rank0 + 2u * popcount_upto(spx_pass_mask(&ps), l);
                        SPX_REQUIRE((size_t) rb.desc_off + rank + 1 < s.descs.size(), "descriptor range");
                        int64_t r, c;
                        uint32_t slot;
                        unit_lane(s, rb, ps, l, r, c, &slot);
                        if (slot != SPX_NO_SLOT) continue;
                        for (uint32_t w = 0; w < ps.width; ++w)
                            if (c + w >= 0 && (size_t)(c + w) < nrows) reached[(size_t)(c + w)] = 1;
                    }
                }
            }
            for (const SpxRowBlock &rb : s.rbs) {
                if (!(rb.flags & SPX_RB_PRIVATE)) continue;
                SPX_REQUIRE(!(rb.flags & SPX_RB_SHARED), "a row-block that stores its rows holds part of an over-long row");
                for (uint32_t r = 0; r < rb.n_rows; ++r)
                    SPX_REQUIRE((size_t) rb.row0 + r < nrows && !reached[(size_t) rb.row0 + r],
                                "a row-block that stores its rows is added to by another one");
            }
        }
    }
    SPX_REQUIRE(s.lds_doubles <= SPX_MAX_WIDE_SLOTS + SPX_MAX_WIDE_ROWS + SPX_MAX_XWIN, "LDS budget");
    SPX_REQUIRE(s.n_spill == 0 || (s.fix_ptr.size() == nrows + 1 && s.fix_idx.size() == s.n_spill),
                "spill lists");
    for (size_t i = 0; i + 1 < s.fix_ptr.size(); ++i)
        SPX_REQUIRE(s.fix_ptr[i] <= s.fix_ptr[i + 1], "spill list order");
    SPX_REQUIRE(s.fix_ptr.empty() || s.fix_ptr.back() == s.fix_idx.size(), "spill list end");
    for (uint32_t k : s.fix_idx) SPX_REQUIRE(k < s.n_spill, "spill index");
    for (uint32_t c : s.spill_col) SPX_REQUIRE(c < nrows, "spill columns");
    SPX_REQUIRE(s.mirror_rows.empty() ? s.mirror_ptr.size() <= 1 && s.mirror_col.empty()
                                      : s.mirror_ptr.size() == s.mirror_rows.size() + 1 &&
                                        s.mirror_ptr.back() == s.mirror_col.size(), "mirror list sizes");
    SPX_REQUIRE(s.mirror_val.empty() || s.mirror_val.size() == s.mirror_col.size(), "mirror list values");
    for (size_t i = 0; i < s.mirror_rows.size(); ++i)
        SPX_REQUIRE(s.mirror_rows[i] < nrows && s.mirror_ptr[i] <= s.mirror_ptr[i + 1] &&
                    (i == 0 || s.mirror_rows[i] > s.mirror_rows[i - 1]), "mirror list rows");
    for (uint32_t c : s.mirror_col) SPX_REQUIRE(c < ncols, "mirror list column");
    SPX_REQUIRE(s.slot_group_col.size() * 8 == s.n_spill, "slot groups");
    for (uint32_t c : s.slot_group_col) SPX_REQUIRE(c % 8 == 0 && (size_t) c + 8 <= nrows, "slot group column");
    SPX_REQUIRE(s.dvalues.empty() || s.dvalues.size() == nrows, "diagonal length");
    bool tiles = false;
    for (size_t i = 0; i < s.rbs.size(); ++i) {
        const SpxRowBlock &rb = s.rbs[i];
        SPX_REQUIRE(rb.n_rows >= 1 && rb.n_rows <= SPX_MAX_WIDE_ROWS, "row-block rows");
        SPX_REQUIRE((size_t) rb.row0 + rb.n_rows <= nrows, "row-block row range");
        SPX_REQUIRE(rb.pass_off == i * (size_t) s.pass_stride && rb.n_pass <= s.pass_stride,
                    "row-block pass range");
        SPX_REQUIRE(rb.val_off <= n_values && rb.val_off % 2 == 0, "row-block value offset");
        SPX_REQUIRE(rb.desc_off <= s.descs.size(), "row-block descriptor offset");
        SPX_REQUIRE((size_t) rb.cidx_off * 16u <= s.cidx.size(), "row-block column-offset position");
        SPX_REQUIRE(rb.cidx_width >= 2 && rb.cidx_width <= 4, "column-offset width");
        SPX_REQUIRE(rb.seg_off <= s.segrows.size(), "row-block row-piece offset");
        SPX_REQUIRE((size_t) rb.n_slots + rb.n_rows + rb.xwin_len <= s.lds_doubles, "row-block LDS use");
        SPX_REQUIRE(rb.xwin_len <= SPX_MAX_XWIN && (size_t) rb.xwin_base + rb.xwin_len <= ncols, "x window");
        SPX_REQUIRE(rb.n_slots == 0 || ((size_t) rb.spill_off + rb.n_slots <= s.n_spill &&
                                        rb.spill_off % 8 == 0 && rb.n_slots % 8 == 0), "spill range");
        if (rb.flags & SPX_RB_SHARED)
            SPX_REQUIRE(rb.carry_slot < s.n_carry && rb.n_rows == 1, "carry slot");
        for (uint32_t t = 0; t < rb.n_pass; ++t) {
            const SpxPass &ps = s.passes[(size_t) rb.pass_off + t];
            const uint32_t nseg = ps.nseg, W = ps.width;
            SPX_REQUIRE(nseg >= 1 && nseg <= SPX_PASS_SEGS && W >= 1 && W <= SPX_MAX_SEG_WIDTH,
                        "pass shape");
            SPX_REQUIRE(ps.val_off % 2 == 0 &&
                        (size_t) rb.val_off + ps.val_off + (size_t) nseg * W <= n_values,
                        "pass value range");
            if (is_gather(ps)) {
                const bool lds = ps.kind == SPX_PASS_GATHER_LDS;
                SPX_REQUIRE((size_t) rb.seg_off + ps.seg0 + nseg <= s.segrows.size(), "row-piece range");
                SPX_REQUIRE(((size_t) rb.cidx_off + (lds ? rb.near_off : 0u)) * 16u +
                            ((size_t) ps.elem0 + (size_t) nseg * W) * (lds || rb.cidx_width == 3 ? 2u : rb.cidx_width)
                                <= s.cidx.size(), "column-offset range");
                SPX_REQUIRE(lds || rb.cidx_width != 3 ||
                            ((size_t) rb.cidx_off + rb.hi_off) * 16u + (size_t) ps.elem0 + (size_t) nseg * W <= s.cidx.size(),
                            "column-offset high bytes");
                for (uint32_t l = 0; l < nseg; ++l) {
                    const uint32_t sr = s.segrows[(size_t) rb.seg_off + ps.seg0 + l];
                    SPX_REQUIRE(SPX_SEGROW_ROW(sr) < rb.n_rows && SPX_SEGROW_LEN(sr) <= W && (sr >> 14) == 0, "row piece row");
                    for (uint32_t w = 0; w < W; ++w) {
                        const int64_t c = gather_col(s, rb, ps, (size_t) ps.elem0 + l + (size_t) w * nseg);
                        SPX_REQUIRE(c >= 0 && (size_t) c < ncols, "gathered column");
                        SPX_REQUIRE(!lds || c < (int64_t) rb.xwin_base + rb.xwin_len, "column outside the x window");
                    }
                }
            } else if (ps.kind == SPX_PASS_SYMTILE) {
                tiles = true;
                SPX_REQUIRE(W == 8 && nseg % 8 == 0, "tile pass shape");
                SPX_REQUIRE((size_t) rb.desc_off + ps.rank0 + nseg / 8 <= s.descs.size(),
                            "tile descriptor range");
                for (uint32_t k = 0; k < nseg / 8; ++k) {
                    const SpxUnitDesc &d = s.descs[(size_t) rb.desc_off + ps.rank0 + k];
                    SPX_REQUIRE(ps.elem0 + (d.bits & 511u) + 8u <= rb.n_rows, "tile rows");
                    SPX_REQUIRE((size_t) d.col0 + 8u <= ncols, "tile columns");
                    SPX_REQUIRE((size_t) (d.bits >> 9) + 8u <= (size_t) rb.n_slots + rb.n_rows, "tile slots");
                }
            } else {
                const bool sym = ps.kind == SPX_PASS_SYMSEG;
                SPX_REQUIRE(ps.kind == SPX_PASS_UNIT || sym, "pass kind");
                SPX_REQUIRE(!sym || s.sym_atomic, "read-once segments without the atomic hand-over");
                SPX_REQUIRE((spx_pass_mask(&ps) & 1ull) == 0, "segment-start mask");
                SPX_REQUIRE(!(ps.flags & SPX_PASSF_INLINE) ||
                            ((size_t) rb.desc_off + ps.rank0 < s.descs.size() &&
                             ps.mask == ((uint64_t) s.descs[(size_t) rb.desc_off + ps.rank0].col0 |
                                         ((uint64_t) s.descs[(size_t) rb.desc_off + ps.rank0].bits << 32))),
                            "inline descriptor");
                const uint32_t last = (uint32_t) ps.rank0 + (sym ? 2u : 1u) * popcount_upto(spx_pass_mask(&ps), nseg - 1) + (sym ? 1u : 0u);
                SPX_REQUIRE((size_t) rb.desc_off + last < s.descs.size(), "descriptor range");
                for (uint32_t l = 0; l < nseg; ++l) {
                    int64_t r, c;
                    uint32_t slot;
                    unit_lane(s, rb, ps, l, r, c, &slot);
                    SPX_REQUIRE(r >= 0 && r < (int64_t) rb.n_rows, "segment row");
                    SPX_REQUIRE(c >= 0 && (size_t) c + W <= ncols, "segment columns");
                    if (sym) {
                        SPX_REQUIRE(c + (int64_t) W <= (int64_t) rb.row0 + r, "read-once segment above the diagonal");
                        SPX_REQUIRE(slot == SPX_NO_SLOT || (size_t) slot + W <= (size_t) rb.n_slots + rb.n_rows,
                                    "segment slots");
                    }
                }
            }
        }
    }
    SPX_REQUIRE(!tiles || s.lds_doubles >= 1, "tile LDS");
    for (const SpxSharedRow &sr : s.shared)
        SPX_REQUIRE(sr.row < nrows && (size_t) sr.first_slot + sr.n_slots <= s.n_carry, "shared row");
#undef SPX_REQUIRE
    return true;
}

}  // namespace spx
