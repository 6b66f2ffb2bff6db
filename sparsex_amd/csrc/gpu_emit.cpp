// gpu_emit.cpp -- see gpu_emit.hpp and gpu_format.h.
#include "gpu_emit.hpp"
#include "threads.hpp"

#include <algorithm>
#include <iterator>
#include <cassert>
#include <cstring>
#include <chrono>

namespace spx {

namespace {

// a part of a unit that lands in one row-block
struct Piece {
    uint32_t elem;     // index of the source unit in Partition::elems
    uint16_t a, b;     // linear / block-col units: element range [a, b)
                       // block-row units: row range [a, b) of the block
};

struct Single { idx_t row, col; val_t val; };   // 0-based row (partition), 0-based col
typedef std::vector<Single, BigAlloc<Single>> SingleVec;   // (the symmetric path holds a triangle's points in these: 6 GB on the contract matrix)
struct RowSeg { idx_t row, col; uint8_t width; val_t v[SPX_MAX_SEG_WIDTH]; };   // same numbering

// Sorts by an integer key first and by `within` among equal keys.  The keys of a row-block's or a
// partition's points (their rows, their columns) span few values compared with their number, so a
// counting pass over the keys and a short sort inside every bucket replaces the comparison sort
// (whose quicksort also degenerates on the runs that mined units leave).  Where (key, within) is a
// total order -- coordinates are unique -- the result is the one std::sort gives.
template <class V, class KeyOf, class Within>
void sort_by_key_then(V &v, KeyOf key_of, Within within)
{
    typedef typename V::value_type T;
    const size_t n = v.size();
    if (n < 2) return;
    auto less = [&](const T &a, const T &b) {
        return key_of(a) < key_of(b) || (key_of(a) == key_of(b) && within(a, b));
    };
    int64_t lo = key_of(v[0]), hi = lo;
    bool sorted = true;
    for (size_t i = 1; i < n; ++i) {
        const int64_t r = key_of(v[i]);
        lo = std::min(lo, r);
        hi = std::max(hi, r);
        sorted = sorted && !less(v[i], v[i - 1]);
    }
    if (sorted) return;
    const uint64_t span = (uint64_t) (hi - lo) + 1;
    if (n < 64 || span > 4 * (uint64_t) n || n > 0xffffffffull) {
        std::sort(v.begin(), v.end(), less);
        return;
    }
    std::vector<uint32_t> start(span + 1, 0);
    for (const T &e : v) ++start[(size_t) (key_of(e) - lo) + 1];
    for (size_t r = 0; r < span; ++r) start[r + 1] += start[r];
    V out(n);
    {
        std::vector<uint32_t> pos(start.begin(), start.end() - 1);
        for (const T &e : v) out[pos[(size_t) (key_of(e) - lo)]++] = e;
    }
    for (size_t r = 0; r < span; ++r) {
        const size_t a = start[r], b = start[r + 1];
        if (b - a < 2) continue;
        if (b - a > 24) { std::sort(out.begin() + a, out.begin() + b, within); continue; }
        for (size_t i = a + 1; i < b; ++i) {
            if (!within(out[i], out[i - 1])) continue;
            T t = out[i];
            size_t j = i;
            for (; j > a && within(t, out[j - 1]); --j) out[j] = out[j - 1];
            out[j] = t;
        }
    }
    v.swap(out);
}

template <class V, class RowOf, class ColOf>
void sort_by_row_col(V &v, RowOf row_of, ColOf col_of)
{
    typedef typename V::value_type T;
    sort_by_key_then(v, row_of, [&](const T &a, const T &b) { return col_of(a) < col_of(b); });
}

inline void sort_singles(SingleVec &v)
{
    sort_by_row_col(v, [](const Single &s) { return (int64_t) s.row; }, [](const Single &s) { return s.col; });
}

struct Plan {
    idx_t row_lo, row_hi;   // rows [lo, hi) of the partition
    bool split;             // one over-long row, chunked
};

template <class V>
inline void pad_to(V &v, size_t mult)
{
    while (v.size() % mult) v.push_back(0.0);
}

// a run of equally wide row segments, values segment-major [nseg][width]
struct Group {
    uint16_t row0;       // relative to the row-block
    uint32_t col0;
    uint16_t nseg;
    uint8_t width;
    uint8_t kind;        // SPX_KIND_*
    uint8_t step;
    uint32_t voff;       // into RbBuilder::gvals_
    uint32_t slot0 = SPX_NO_SLOT;   // SPX_PASS_SYMSEG: slot of segment 0's first column
    bool pure = false;              // read-once segments, spx.gpu.sym_pure_passes: fills passes of its own (split_pure_groups)
};

// (read-once passes of one unit: a run of this many segments and more gets passes of its own)
constexpr uint32_t PURE_MIN_SEGS = 40;      // (a pass of its own is at least five eighths full)

class RbBuilder {
public:
    RbBuilder(const Partition &p, GpuStream &out, bool stack = true, bool x_window = true, bool inline_desc = true,
              bool pure_passes = false)
        : p_(p), out_(out), stack_(stack), x_window_(x_window), inline_desc_(inline_desc), pure_passes_(pure_passes) {}

    // rows [lo, hi) of the partition with what the planner cut out for them
    struct Part {
        idx_t lo, hi;
        const std::vector<Piece> *pieces;
        SingleVec *singles;
        const std::vector<const SymTile *> *tiles;
        const std::vector<RowSeg> *rowsegs;
        const std::vector<const SymSeg *> *symsegs;
    };
    // emits one row-block for rows [lo, hi) from the given pieces/singles
    void emit(idx_t lo, idx_t hi, const std::vector<Piece> &pieces,
              SingleVec &singles, uint8_t flags, uint32_t carry_slot,
              const std::vector<const SymTile *> *tiles = nullptr,
              const std::vector<RowSeg> *rowsegs = nullptr,
              const std::vector<const SymSeg *> *symsegs = nullptr)
    {
        std::vector<Part> one{Part{lo, hi, &pieces, &singles, tiles, rowsegs, symsegs}};
        emit(one, flags, carry_slot);
    }
    // ... or one wide row-block from consecutive parts (each at most SPX_MAX_RB_ROWS rows,
    // SPX_MAX_WIDE_ROWS together): one y tile, one set of transposed-sum slots
    void emit(std::vector<Part> &parts, uint8_t flags, uint32_t carry_slot);

private:
    void add_group(idx_t row, idx_t col, size_t nseg, size_t width, unsigned kind, unsigned step)
    {
        Group g;
        g.row0 = (uint16_t) row;
        g.col0 = (uint32_t) col;
        g.nseg = (uint16_t) nseg;
        g.width = (uint8_t) width;
        g.kind = (uint8_t) kind;
        g.step = (uint8_t) step;
        g.voff = (uint32_t) gvals_.size();
        groups_.push_back(g);
    }
    void groups_from_piece(const Piece &pc, idx_t lo);
    void stack_groups();
    // which groups a call of emit_unit_passes takes: all of them, or (read-once segments with passes of their own)
    // only the groups marked `pure`, each filling passes that hold nothing else / only the others
    enum PassSet { ALL_GROUPS, PURE_GROUPS, MIXED_GROUPS };
    void emit_unit_passes(SpxRowBlock &rb, bool sym = false, uint32_t row_base = 0, PassSet which = ALL_GROUPS);
    void split_pure_groups();
    void assign_slots(SpxRowBlock &rb, const std::vector<const SymTile *> *tiles,
                      const std::vector<const SymSeg *> *symsegs);
    void slot_symseg_groups(const SpxRowBlock &rb);
    // slot of global column c in the current row-block (SPX_NO_SLOT: none)
    uint32_t slot_of(const SpxRowBlock &rb, idx_t c) const
    {
        if (c >= (idx_t) rb.row0) return (uint32_t) rb.n_slots + (uint32_t)(c - (idx_t) rb.row0);
        const idx_t g = c & ~(idx_t) 7;
        auto it = std::lower_bound(slot_groups_.begin(), slot_groups_.end(), g);
        if (it == slot_groups_.end() || *it != g) return SPX_NO_SLOT;
        return (uint32_t)(it - slot_groups_.begin()) * 8u + (uint32_t)(c & 7);
    }
    void emit_gather_passes(SpxRowBlock &rb, SingleVec &singles, idx_t lo);
    void emit_tile_passes(SpxRowBlock &rb, const std::vector<const SymTile *> &tiles, uint32_t row_base = 0);

    const Partition &p_;
    GpuStream &out_;
    bool stack_;
    bool x_window_;
    bool inline_desc_;
    bool pure_passes_;
    std::vector<Group> groups_;
    std::vector<val_t> gvals_;
    std::vector<idx_t> slot_groups_;   // first columns of the row-block's slot groups (ascending)
};

void RbBuilder::groups_from_piece(const Piece &pc, idx_t lo)
{
    const Elem &u = p_.elems[pc.elem];
    const val_t *src = &p_.pool[u.voff];
    const size_t WMAX = SPX_MAX_SEG_WIDTH;
    if (enc_is_block_row(u.type)) {
        // rows [a,b) of an R x cdim column-major block, emitted row-major in
        // column chunks of at most WMAX
        const size_t R = (size_t) enc_block_align(u.type);
        const size_t cdim = u.size / R;
        const size_t rr = pc.b - pc.a;
        for (size_t c0 = 0; c0 < cdim; c0 += WMAX) {
            size_t w = std::min(WMAX, cdim - c0);
            add_group(u.row - 1 + (idx_t) pc.a - lo, u.col - 1 + (idx_t) c0, rr, w, SPX_KIND_BLOCK, 0);
            for (size_t s = 0; s < rr; ++s)
                for (size_t i = 0; i < w; ++i)
                    gvals_.push_back(src[(c0 + i) * R + (pc.a + s)]);
        }
        return;
    }
    idx_t r0, c0;
    unit_elem_coords(u, pc.a, r0, c0);
    const size_t n = pc.b - pc.a;
    if (enc_is_block_col(u.type)) {
        // whole rows of a rdim x C row-major block (cuts fall on row borders)
        const size_t C = (size_t) enc_block_align(u.type);
        const size_t rr = n / C;
        for (size_t cc = 0; cc < C; cc += WMAX) {
            size_t w = std::min(WMAX, C - cc);
            add_group(r0 - 1 - lo, c0 - 1 + (idx_t) cc, rr, w, SPX_KIND_BLOCK, 0);
            for (size_t s = 0; s < rr; ++s)
                for (size_t i = 0; i < w; ++i)
                    gvals_.push_back(src[pc.a + s * C + cc + i]);
        }
        return;
    }
    const int d = (int) u.delta;
    if (u.type == ENC_H && d == 1) {
        const size_t CH = SPX_HORIZ_CHUNK;
        const size_t nf = n / CH, m = n % CH;
        if (nf) {
            add_group(r0 - 1 - lo, c0 - 1, nf, CH, SPX_KIND_HORIZ, (unsigned) CH);
            gvals_.insert(gvals_.end(), src + pc.a, src + pc.a + nf * CH);
        }
        if (m) {
            add_group(r0 - 1 - lo, c0 - 1 + (idx_t)(nf * CH), 1, m, SPX_KIND_HORIZ, 0);
            gvals_.insert(gvals_.end(), src + pc.a + nf * CH, src + pc.b);
        }
        return;
    }
    unsigned kind = SPX_KIND_HORIZ;
    switch (u.type) {
    case ENC_H: kind = SPX_KIND_HORIZ; break;
    case ENC_V: kind = SPX_KIND_VERT; break;
    case ENC_D: kind = SPX_KIND_DIAG; break;
    case ENC_AD: kind = SPX_KIND_ADIAG; break;
    default: assert(false);
    }
    assert((unsigned) d <= SPX_MAX_STEP);
    add_group(r0 - 1 - lo, c0 - 1, n, 1, kind, (unsigned) d);
    gvals_.insert(gvals_.end(), src + pc.a, src + pc.b);
}

// Equal row segments that follow a regular course share one descriptor,
// whatever units they came from (the pieces of a cut block, horizontal units of
// neighbouring rows, the chunks of a wide horizontal unit, re-cut stencil rows):
//   * along a diagonal -- same width, rows and first columns advancing by the
//     same step (a stencil; the 3x3 node blocks of an FE matrix, row by row);
//     long chains (>= 8 segments) are taken first,
//   * on top of each other -- same columns, consecutive rows: a dense block,
//   * next to each other in one row (full chunks),
//   * short diagonal chains among what is left.
// Fewer descriptors, same lanes.
void RbBuilder::stack_groups()
{
    struct Run { uint16_t row0; uint32_t col0; uint32_t voff; uint8_t width; };   // one row segment
    std::vector<Run> runs;
    std::vector<Group> kept;
    for (const Group &g : groups_) {
        if (g.kind == SPX_KIND_BLOCK || g.nseg == 1) {
            for (uint32_t s = 0; s < g.nseg; ++s)
                runs.push_back(Run{(uint16_t)(g.row0 + s), g.col0, g.voff + s * g.width, g.width});
        } else if (g.kind == SPX_KIND_HORIZ && g.step == SPX_HORIZ_CHUNK && g.width == SPX_HORIZ_CHUNK) {
            for (uint32_t s = 0; s < g.nseg; ++s)
                runs.push_back(Run{g.row0, g.col0 + s * SPX_HORIZ_CHUNK, g.voff + s * SPX_HORIZ_CHUNK, g.width});
        } else {
            kept.push_back(g);
        }
    }
    if (runs.empty()) return;
    std::vector<val_t> vals;
    vals.reserve(gvals_.size());
    for (Group &g : kept) {
        const uint32_t off = (uint32_t) vals.size();
        vals.insert(vals.end(), gvals_.begin() + g.voff, gvals_.begin() + g.voff + (size_t) g.nseg * g.width);
        g.voff = off;
    }
    // emits runs[idx[i..j)] as one group
    auto emit_chain = [&](const std::vector<uint32_t> &idx, size_t i, size_t j, unsigned kind, unsigned step) {
        const Run &r0 = runs[idx[i]];
        Group g;
        g.row0 = r0.row0;
        g.col0 = r0.col0;
        g.nseg = (uint16_t)(j - i);
        g.width = r0.width;
        g.kind = (uint8_t) kind;
        g.step = (uint8_t) step;
        g.voff = (uint32_t) vals.size();
        for (size_t k = i; k < j; ++k)
            vals.insert(vals.end(), gvals_.begin() + runs[idx[k]].voff,
                        gvals_.begin() + runs[idx[k]].voff + g.width);
        kept.push_back(g);
    };
    auto diag_id = [&](uint32_t k) { return (int64_t) runs[k].col0 - (int64_t) runs[k].row0; };
    // index lists are sorted through one packed 64-bit key per run (width: 8 bits, a column or a
    // diagonal: 33, the row inside the row-block: 16) instead of a comparison that looks three
    // fields up per step; the keys are unique (no two runs start on the same nonzero)
    struct Keyed { uint64_t key; uint32_t idx; };
    std::vector<Keyed> keyed;
    auto sort_by_key = [&](std::vector<uint32_t> &idx, auto key_of) {
        keyed.resize(idx.size());
        for (size_t i = 0; i < idx.size(); ++i) keyed[i] = Keyed{key_of(idx[i]), idx[i]};
        std::sort(keyed.begin(), keyed.end(), [](const Keyed &a, const Keyed &b) { return a.key < b.key; });
        for (size_t i = 0; i < idx.size(); ++i) idx[i] = keyed[i].idx;
    };
    // chains along diagonals; `min_len` segments at least; returns what is left
    auto diagonal_chains = [&](std::vector<uint32_t> idx, size_t min_len) {
        sort_by_key(idx, [&](uint32_t k) {
            return (uint64_t) runs[k].width << 56 | (uint64_t) (diag_id(k) + 65536) << 16 | runs[k].row0;
        });
        std::vector<uint32_t> left;
        for (size_t i = 0; i < idx.size();) {
            size_t j = i + 1;
            unsigned step = 0;
            auto same = [&](size_t k) {
                return runs[idx[k]].width == runs[idx[i]].width && diag_id(idx[k]) == diag_id(idx[i]);
            };
            if (j < idx.size() && same(j) &&
                (unsigned)(runs[idx[j]].row0 - runs[idx[i]].row0) <= SPX_MAX_STEP) {
                step = (unsigned)(runs[idx[j]].row0 - runs[idx[i]].row0);
                while (j < idx.size() && same(j) &&
                       runs[idx[j]].row0 == runs[idx[i]].row0 + (j - i) * step)
                    ++j;
            }
            if (step && j - i >= min_len) {
                emit_chain(idx, i, j, SPX_KIND_DIAG, step);
                i = j;
            } else {
                left.push_back(idx[i]);
                ++i;
            }
        }
        return left;
    };
    std::vector<uint32_t> idx(runs.size());
    for (uint32_t k = 0; k < idx.size(); ++k) idx[k] = k;
    idx = diagonal_chains(idx, 8);
    // dense blocks
    sort_by_key(idx, [&](uint32_t k) {
        return (uint64_t) runs[k].width << 56 | (uint64_t) runs[k].col0 << 16 | runs[k].row0;
    });
    std::vector<uint32_t> lone;
    for (size_t i = 0; i < idx.size();) {
        size_t j = i + 1;
        while (j < idx.size() && runs[idx[j]].width == runs[idx[i]].width &&
               runs[idx[j]].col0 == runs[idx[i]].col0 &&
               runs[idx[j]].row0 == runs[idx[i]].row0 + (j - i))
            ++j;
        if (j - i > 1) emit_chain(idx, i, j, SPX_KIND_BLOCK, 0);
        else lone.push_back(idx[i]);
        i = j;
    }
    // full chunks next to each other in one row
    sort_by_key(lone, [&](uint32_t k) { return (uint64_t) runs[k].row0 << 32 | runs[k].col0; });
    std::vector<uint32_t> rest;
    for (size_t i = 0; i < lone.size();) {
        size_t j = i + 1;
        if (runs[lone[i]].width == SPX_HORIZ_CHUNK)
            while (j < lone.size() && runs[lone[j]].width == SPX_HORIZ_CHUNK &&
                   runs[lone[j]].row0 == runs[lone[i]].row0 &&
                   runs[lone[j]].col0 == runs[lone[i]].col0 + (uint32_t)((j - i) * SPX_HORIZ_CHUNK))
                ++j;
        if (j - i > 1) emit_chain(lone, i, j, SPX_KIND_HORIZ, SPX_HORIZ_CHUNK);
        else rest.push_back(lone[i]);
        i = j;
    }
    // short diagonal chains, then segments of their own
    rest = diagonal_chains(rest, 2);
    for (size_t i = 0; i < rest.size(); ++i) emit_chain(rest, i, i + 1, SPX_KIND_HORIZ, 0);
    groups_.swap(kept);
    gvals_.swap(vals);
}

// Read-once segments, spx.gpu.sym_pure_passes: a long run of segments (a diagonal chain of a stencil, the rows
// of a block) is given passes that hold nothing but its own lanes.  Such a pass has ONE descriptor, which
// travels in its header (SPX_PASSF_INLINE): the kernel knows every lane's row, columns and slot from the
// header alone and requests x together with the values (csx_spmv_sx_kernel), instead of after a descriptor
// load.  The last pass of a run may be partly filled -- idle lanes cost no bytes -- but a tail shorter than
// PURE_MIN_SEGS is cut off as a unit of its own (8 + 8 bytes of descriptor) and shares passes with the
// other short ones.
void RbBuilder::split_pure_groups()
{
    std::vector<Group> out;
    out.reserve(groups_.size() + 16);
    for (Group g : groups_) {
        if (g.nseg < PURE_MIN_SEGS) {
            out.push_back(g);
            continue;
        }
        const uint32_t tail = g.nseg % SPX_PASS_SEGS;
        if (tail == 0 || tail >= PURE_MIN_SEGS) {
            g.pure = true;
            out.push_back(g);
            continue;
        }
        const int dcol = (g.kind == SPX_KIND_HORIZ || g.kind == SPX_KIND_DIAG) ? (int) g.step
                         : (g.kind == SPX_KIND_ADIAG ? -(int) g.step : 0);
        const int drow = g.kind == SPX_KIND_BLOCK ? 1 : (g.kind >= SPX_KIND_VERT ? (int) g.step : 0);
        const uint32_t head = g.nseg - tail;
        Group t = g;
        t.row0 = (uint16_t)(g.row0 + head * (uint32_t) drow);
        t.col0 = (uint32_t)((int64_t) g.col0 + (int64_t) head * dcol);
        t.nseg = (uint16_t) tail;
        t.voff = g.voff + head * g.width;
        if (g.slot0 != SPX_NO_SLOT) t.slot0 = (uint32_t)((int64_t) g.slot0 + (int64_t) head * dcol);
        g.nseg = (uint16_t) head;
        g.pure = true;
        out.push_back(g);
        out.push_back(t);
    }
    groups_.swap(out);
}

void RbBuilder::emit_unit_passes(SpxRowBlock &rb, bool sym, uint32_t row_base, PassSet which)
{
    // passes hold segments of one width: order the groups by width
    std::vector<uint32_t> order(groups_.size());
    for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        return groups_[a].width < groups_[b].width;
    });
    struct Slot { uint32_t desc; uint32_t grp; uint16_t s; };
    std::vector<Slot> lanes;
    uint32_t seg_counter = 0;
    auto flush = [&](uint8_t width) {
        if (lanes.empty()) return;
        SpxPass ps;
        std::memset(&ps, 0, sizeof(ps));
        if (out_.values.size() % 2) out_.values.push_back(0.0);
        ps.val_off = (uint32_t)(out_.values.size() - rb.val_off);
        ps.rank0 = (uint16_t) lanes[0].desc;
        ps.seg0 = (uint16_t)(seg_counter - lanes.size());
        ps.nseg = (uint8_t) lanes.size();
        ps.width = width;
        ps.kind = sym ? SPX_PASS_SYMSEG : SPX_PASS_UNIT;
        ps.elem0 = row_base;
        const size_t nseg = lanes.size();
        size_t base = out_.values.size();
        out_.values.resize(base + nseg * width, 0.0);
        for (size_t l = 0; l < nseg; ++l) {
            if (l > 0 && lanes[l].desc != lanes[l - 1].desc) ps.mask |= 1ull << l;
            const Group &g = groups_[lanes[l].grp];
            for (uint32_t w = 0; w < width; ++w)
                out_.values[base + spx_pass_value_index((uint32_t) l, w, (uint32_t) nseg, width)] =
                    gvals_[g.voff + (size_t) lanes[l].s * width + w];
        }
        if (ps.mask == 0 && inline_desc_) {
            // one descriptor for the whole pass: it travels in the pass header (SPX_PASSF_INLINE;
            // a read-once pass still fetches its slot entry, which it needs last)
            const SpxUnitDesc &d = out_.descs[(size_t) rb.desc_off + ps.rank0];
            ps.mask = (uint64_t) d.col0 | ((uint64_t) d.bits << 32);
            ps.flags |= SPX_PASSF_INLINE;
        }
        out_.passes.push_back(ps);
        ++rb.n_pass;
        lanes.clear();
    };
    uint8_t cur_w = 0;
    for (uint32_t gi : order) {
        const Group &g = groups_[gi];
        if (which == PURE_GROUPS && !g.pure) continue;
        if (which == MIXED_GROUPS && g.pure) continue;
        if (g.width != cur_w || which == PURE_GROUPS) {
            flush(cur_w);
            cur_w = g.width;
        }
        SpxUnitDesc d;
        d.col0 = g.col0;
        d.bits = spx_desc_bits(g.row0, seg_counter, g.kind, g.step);
        uint32_t di = (uint32_t)(out_.descs.size() - rb.desc_off);
        out_.descs.push_back(d);
        if (sym) {
            SpxUnitDesc d2;        // second half of a symmetric unit's descriptor: its slot
            d2.col0 = g.slot0;
            d2.bits = 0;
            out_.descs.push_back(d2);
        }
        ++out_.n_units;
        for (uint16_t s = 0; s < g.nseg; ++s) {
            lanes.push_back(Slot{di, gi, s});
            ++seg_counter;
            if (lanes.size() == SPX_PASS_SEGS) flush(cur_w);
        }
    }
    flush(cur_w);
}

// Leftover nonzeros as row pieces: every row's leftovers are cut into pieces of
// at most SPX_MAX_SEG_WIDTH nonzeros; a lane owns one piece (its values, its
// column offsets) and adds ONE partial sum to the y tile -- no cross-lane
// reduction, and lanes of a pass mostly hit different rows (same-address LDS
// adds are serialised at ~3 clocks each).  Pieces are grouped by size so that a
// pass is uniform, like the unit passes.
//
// Where many of the row-block's leftovers have their columns close together (a
// web graph's links inside a site, the couplings of a band), those columns
// become the row-block's x window: the workgroup stages x[window] in LDS with
// coalesced loads and these leftovers gather from LDS (SPX_PASS_GATHER_LDS, u16
// offsets); the others gather through L2 as before.
void RbBuilder::emit_gather_passes(SpxRowBlock &rb, SingleVec &singles, idx_t lo)
{
    sort_singles(singles);
    const size_t n = singles.size();
    rb.seg_off = (uint32_t) out_.segrows.size();
    while (out_.cidx.size() % 16) out_.cidx.push_back(0);
    const size_t cbytes = out_.cidx.size();           // 16-byte aligned
    rb.cidx_off = (uint32_t)(cbytes / 16);
    rb.cidx_width = 2;
    if (!n) return;

    // The x window: the run of columns (at most SPX_MAX_XWIN wide) whose staging
    // pays best.  Measured on syn-webbase (profiles/r02/ablation.md): leftovers near
    // the diagonal already hit L2, so a sparse window (one leftover per four
    // doubles staged) only adds passes, 39.2 -> 39.9 us; staging pays where the
    // window is dense -- a band that CSX left unencoded.  Maximise 2 * count - span.
    idx_t wlo = 0, whi = -1;
    if (x_window_ && n >= 64) {
        std::vector<idx_t> cols(n);
        for (size_t i = 0; i < n; ++i) cols[i] = singles[i].col;
        std::sort(cols.begin(), cols.end());
        // for the right end b the best left end a maximises cols[a] - 2a among the
        // a with cols[b] - cols[a] < SPX_MAX_XWIN - 1 (a monotonic queue)
        std::vector<size_t> dq;
        size_t head = 0;
        int64_t best = 0;
        auto key = [&](size_t a) { return (int64_t) cols[a] - 2 * (int64_t) a; };
        for (size_t b = 0; b < n; ++b) {
            while (dq.size() > head && key(dq.back()) <= key(b)) dq.pop_back();
            dq.push_back(b);
            while (cols[b] - cols[dq[head]] >= (idx_t) SPX_MAX_XWIN - 1) ++head;
            const size_t a = dq[head];
            const int64_t gain = 2 * (int64_t)(b - a + 1) - (int64_t)(cols[b] - cols[a] + 1);
            if (gain > best) {
                best = gain;
                wlo = cols[a];
                whi = cols[b];
            }
        }
        wlo &= ~(idx_t) 1;
        size_t inside = 0;
        for (idx_t c : cols) inside += c >= wlo && c <= whi;
        if (inside < 64 || best <= 0) whi = wlo - 1;     // too few to be worth a pass of their own
    }
    SingleVec near;
    if (whi >= wlo) {
        SingleVec far;
        for (const Single &s : singles) (s.col >= wlo && s.col <= whi ? near : far).push_back(s);
        singles.swap(far);
        rb.xwin_base = (uint32_t) wlo;
        rb.xwin_len = (uint16_t)(whi - wlo + 1);
        out_.lds_doubles = std::max<uint32_t>(out_.lds_doubles,
                                              (uint32_t) rb.n_slots + rb.n_rows + rb.xwin_len);
    }

    struct Piece2 { uint32_t first; uint8_t width; };
    size_t pieces_before = 0;
    // one set of passes: `set` sorted by (row, col); offsets of `width` bytes relative
    // to `base`, appended to the row-block's offset area.  Pieces are taken in order
    // of their size, 64 to a pass; a pass is as wide as its longest piece and the
    // shorter ones are padded (zero value, offset 0; the piece's length travels with
    // its row in `segrows`, the lanes skip what is not there).  A pass is closed
    // early when the padding would pass 30 % of it.
    auto emit_set = [&](const SingleVec &set, uint8_t kind, idx_t base, unsigned width,
                        size_t area) {
        const size_t m = set.size();
        std::vector<Piece2> pcs;
        for (size_t i = 0; i < m;) {
            size_t j = i;
            while (j < m && set[j].row == set[i].row) ++j;
            for (size_t k = i; k < j; k += SPX_MAX_SEG_WIDTH)
                pcs.push_back(Piece2{(uint32_t) k, (uint8_t) std::min<size_t>(SPX_MAX_SEG_WIDTH, j - k)});
            i = j;
        }
        std::stable_sort(pcs.begin(), pcs.end(),
                         [](const Piece2 &a, const Piece2 &b) { return a.width < b.width; });
        // where one pass ends (the same rule below): needed up front for 3-byte offsets,
        // whose high bytes form an array of their own behind all the low halves
        auto pass_end = [&](size_t b) {
            size_t e = b, real = 0;
            while (e < pcs.size() && e - b < SPX_PASS_SEGS) {
                const size_t w = pcs[e].width, lanes = e - b + 1;
                if (e - b >= 16 && (lanes * w - (real + w)) * 10 > lanes * w * 3) break;
                real += w;
                ++e;
            }
            return e;
        };
        size_t slots_total = 0;
        for (size_t b = 0; b < pcs.size();) {
            const size_t e = pass_end(b);
            slots_total += (e - b) * pcs[e - 1].width;
            b = e;
        }
        size_t hi_area = 0;
        if (width == 3) {
            hi_area = (area + slots_total * 2 + 15) / 16 * 16;
            rb.hi_off = (uint32_t)((hi_area - cbytes) / 16);
            out_.cidx.resize(hi_area + slots_total, 0);
        }
        auto put_off = [&](size_t at, uint32_t off) {
            if (width == 2) {
                uint16_t o = (uint16_t) off;
                std::memcpy(&out_.cidx[area + at * 2], &o, 2);
            } else if (width == 3) {
                uint16_t o = (uint16_t) off;
                std::memcpy(&out_.cidx[area + at * 2], &o, 2);
                out_.cidx[hi_area + at] = (uint8_t)(off >> 16);
            } else {
                std::memcpy(&out_.cidx[area + at * 4], &off, 4);
            }
        };
        size_t elems_before = 0;
        for (size_t b = 0; b < pcs.size();) {
            const size_t e = pass_end(b);
            const uint32_t W = pcs[e - 1].width;
            const size_t nseg = e - b;
            SpxPass ps;
            std::memset(&ps, 0, sizeof(ps));
            if (out_.values.size() % 2) out_.values.push_back(0.0);
            ps.val_off = (uint32_t)(out_.values.size() - rb.val_off);
            ps.seg0 = (uint16_t) pieces_before;
            ps.nseg = (uint8_t) nseg;
            ps.width = (uint8_t) W;
            ps.kind = kind;
            ps.elem0 = (uint32_t) elems_before;
            const size_t vbase = out_.values.size();
            out_.values.resize(vbase + nseg * W, 0.0);
            if (width != 3) out_.cidx.resize(area + (elems_before + nseg * W) * width, 0);
            for (size_t l = 0; l < nseg; ++l) {
                const Piece2 &pc = pcs[b + l];
                out_.segrows.push_back(SPX_SEGROW((uint32_t)(set[pc.first].row - lo), (uint32_t) pc.width));
                for (uint32_t w = 0; w < pc.width; ++w) {
                    const Single &s = set[pc.first + w];
                    out_.values[vbase + spx_pass_value_index((uint32_t) l, w, (uint32_t) nseg, W)] = s.val;
                    put_off(elems_before + (size_t) w * nseg + l, (uint32_t)(s.col - base));
                }
            }
            out_.passes.push_back(ps);
            ++rb.n_pass;
            elems_before += nseg * W;
            pieces_before += nseg;
            b = e;
        }
    };
    if (!singles.empty()) {
        idx_t cmin = singles[0].col, cmax = singles[0].col;
        for (const Single &s : singles) {
            cmin = std::min(cmin, s.col);
            cmax = std::max(cmax, s.col);
        }
        rb.cbase = (uint32_t) cmin;
        rb.cidx_width = ((size_t)(cmax - cmin) < 65536) ? 2 : ((size_t)(cmax - cmin) < (1u << 24) ? 3 : 4);
        emit_set(singles, SPX_PASS_GATHER, cmin, rb.cidx_width, cbytes);
    }
    if (!near.empty()) {
        while (out_.cidx.size() % 16) out_.cidx.push_back(0);
        const size_t nbytes = out_.cidx.size();
        if ((nbytes - cbytes) / 16 > 0xffff) throw FatalError("offset area of a row-block too large");
        rb.near_off = (uint16_t)((nbytes - cbytes) / 16);
        emit_set(near, SPX_PASS_GATHER_LDS, wlo, 2, nbytes);
        // (the caller accounts for the leftovers through `singles`)
        singles.insert(singles.end(), near.begin(), near.end());
    }
}

// The transposed-sum slots of a row-block: one group of eight per aligned group of eight
// columns in front of its first row that a tile or a read-once row segment touches (tiles
// first: they cannot do without), as many as the LDS holds.
void RbBuilder::assign_slots(SpxRowBlock &rb, const std::vector<const SymTile *> *tiles,
                             const std::vector<const SymSeg *> *symsegs)
{
    const idx_t row0 = (idx_t) rb.row0;
    slot_groups_.clear();
    if (tiles)
        for (const SymTile *t : *tiles)
            if (t->col0 < row0) slot_groups_.push_back(t->col0 & ~(idx_t) 7);
    std::sort(slot_groups_.begin(), slot_groups_.end());
    slot_groups_.erase(std::unique(slot_groups_.begin(), slot_groups_.end()), slot_groups_.end());
    assert(slot_groups_.size() * 8 <= SPX_MAX_TILE_SLOTS);
    const size_t max_slots = rb.n_rows > SPX_MAX_RB_ROWS ? SPX_MAX_WIDE_SLOTS : SPX_MAX_TILE_SLOTS;
    if (symsegs && !symsegs->empty()) {
        std::vector<idx_t> more;
        // (a group is handed to y whole, eight doubles: one that would reach past the last row of
        // this range -- possible only in front of a row-block of its last seven rows -- is left
        // out; its segments add to y themselves)
        const idx_t end = p_.row_start + (idx_t) p_.nr_rows;
        for (const SymSeg *sg : *symsegs)
            for (idx_t c = sg->col & ~(idx_t) 7; c < sg->col + sg->width && c < row0; c += 8)
                if (c + 8 <= end) more.push_back(c);
        std::sort(more.begin(), more.end());
        more.erase(std::unique(more.begin(), more.end()), more.end());
        // (nearest to the row-block first, should they not all fit: the others add to y directly)
        for (size_t k = more.size(); k-- > 0;) {
            if ((slot_groups_.size() + 1) * 8 > max_slots) break;
            if (!std::binary_search(slot_groups_.begin(), slot_groups_.end(), more[k])) slot_groups_.push_back(more[k]);
        }
        std::sort(slot_groups_.begin(), slot_groups_.end());
        slot_groups_.erase(std::unique(slot_groups_.begin(), slot_groups_.end()), slot_groups_.end());
    }
    rb.n_slots = (uint16_t)(slot_groups_.size() * 8);
    rb.spill_off = (uint32_t) out_.spill_col.size();
    for (idx_t g : slot_groups_)
        for (idx_t c = g; c < g + 8; ++c) out_.spill_col.push_back((uint32_t) c);
    out_.lds_doubles = std::max<uint32_t>(out_.lds_doubles, (uint32_t) rb.n_slots + rb.n_rows);
}

// Gives every group of read-once segments (groups_, after stacking) its slot: segment s of a
// group uses slot0 + s * dcol, so the slots of all its segments must lie in one run of
// consecutive slots; a group whose segments do not (they leave a window of touched columns,
// or cross the row-block's first row where the slots do not continue into the y tile) is
// broken into its segments, and a segment that still has no run of slots of its own gets none.
void RbBuilder::slot_symseg_groups(const SpxRowBlock &rb)
{
    auto seg_slot = [&](idx_t col, unsigned width) -> uint32_t {
        const uint32_t s0 = slot_of(rb, col);
        if (s0 == SPX_NO_SLOT) return SPX_NO_SLOT;
        for (unsigned w = 1; w < width; ++w)
            if (slot_of(rb, col + (idx_t) w) != s0 + w) return SPX_NO_SLOT;
        return s0;
    };
    std::vector<Group> out;
    for (Group g : groups_) {
        const int dcol = (g.kind == SPX_KIND_HORIZ || g.kind == SPX_KIND_DIAG) ? (int) g.step
                         : (g.kind == SPX_KIND_ADIAG ? -(int) g.step : 0);
        const int drow = g.kind == SPX_KIND_BLOCK ? 1 : (g.kind >= SPX_KIND_VERT ? (int) g.step : 0);
        const uint32_t s0 = seg_slot((idx_t) g.col0, g.width);
        bool ok = s0 != SPX_NO_SLOT;
        for (uint32_t sidx = 1; ok && sidx < g.nseg; ++sidx)
            ok = seg_slot((idx_t) g.col0 + (idx_t) sidx * dcol, g.width) == s0 + (uint32_t)((int) sidx * dcol);
        if (ok || g.nseg == 1) {
            g.slot0 = s0;
            out.push_back(g);
            continue;
        }
        for (uint32_t sidx = 0; sidx < g.nseg; ++sidx) {
            Group one = g;
            one.row0 = (uint16_t)(g.row0 + sidx * drow);
            one.col0 = (uint32_t)((int64_t) g.col0 + (int64_t) sidx * dcol);
            one.nseg = 1;
            one.kind = SPX_KIND_HORIZ;
            one.step = 0;
            one.voff = g.voff + sidx * g.width;
            one.slot0 = seg_slot((idx_t) one.col0, g.width);
            out.push_back(one);
        }
    }
    for (const Group &g : out) {
        if (g.slot0 != SPX_NO_SLOT) continue;
        const int dcol = (g.kind == SPX_KIND_HORIZ || g.kind == SPX_KIND_DIAG) ? (int) g.step
                         : (g.kind == SPX_KIND_ADIAG ? -(int) g.step : 0);
        for (uint32_t sidx = 0; sidx < g.nseg; ++sidx)
            out_.direct_cols.emplace_back((uint32_t)((int64_t) g.col0 + (int64_t) sidx * dcol), (uint32_t) g.width);
    }
    groups_.swap(out);
}

// Symmetric tiles of this row-block: eight tiles per pass, lanes 8t..8t+7 the
// rows of tile t.  The transposed sums of a tile's eight columns go to
// consecutive slots: columns in front of the row-block's first row are ranked
// (slot = number of touched columns in front), columns inside the row-block
// continue behind them at n_slots + (column - row0), i.e. in the y tile itself.
void RbBuilder::emit_tile_passes(SpxRowBlock &rb, const std::vector<const SymTile *> &tiles, uint32_t row_base)
{
    if (tiles.empty()) return;
    const idx_t row0 = (idx_t) rb.row0 + (idx_t) row_base;     // first row of this part
    // (the slots were laid out by assign_slots: tiles start on columns that are multiples of
    // eight, a group of eight columns that begins in front of the row-block is taken whole)
    for (size_t b = 0; b < tiles.size(); b += 8) {
        const size_t nt = std::min<size_t>(8, tiles.size() - b);
        const uint32_t nseg = (uint32_t)(8 * nt);
        SpxPass ps;
        std::memset(&ps, 0, sizeof(ps));
        if (out_.values.size() % 2) out_.values.push_back(0.0);
        ps.val_off = (uint32_t)(out_.values.size() - rb.val_off);
        ps.rank0 = (uint16_t)(out_.descs.size() - rb.desc_off);
        ps.nseg = (uint8_t) nseg;
        ps.width = 8;
        ps.kind = SPX_PASS_SYMTILE;
        ps.elem0 = row_base;
        const size_t base = out_.values.size();
        out_.values.resize(base + (size_t) nseg * 8, 0.0);
        for (size_t k = 0; k < nt; ++k) {
            const SymTile &t = *tiles[b + k];
            const uint32_t slot = slot_of(rb, t.col0);
            assert(slot != SPX_NO_SLOT);
            SpxUnitDesc d;
            d.col0 = (uint32_t) t.col0;
            d.bits = (uint32_t)(t.row0 - row0) | (slot << 9);
            out_.descs.push_back(d);
            ++out_.n_units;
            for (uint32_t i = 0; i < 8; ++i)
                for (uint32_t w = 0; w < 8; ++w)
                    out_.values[base + spx_pass_value_index((uint32_t)(8 * k) + i, w, nseg, 8)] =
                        t.v[i * 8 + w];
        }
        out_.passes.push_back(ps);
        ++rb.n_pass;
    }
    out_.n_unit_elems += 64 * tiles.size();
    out_.nnz_stored += 64 * tiles.size();
}

void RbBuilder::emit(std::vector<Part> &parts, uint8_t flags, uint32_t carry_slot)
{
    const idx_t lo = parts.front().lo, hi = parts.back().hi;
    assert(parts.size() == 1 ? hi - lo <= (idx_t) SPX_MAX_RB_ROWS : hi - lo <= (idx_t) SPX_MAX_WIDE_ROWS);
    SpxRowBlock rb;
    std::memset(&rb, 0, sizeof(rb));
    pad_to(out_.values, 2);
    rb.val_off = out_.values.size();
    rb.pass_off = (uint32_t) out_.passes.size();
    rb.desc_off = (uint32_t) out_.descs.size();
    rb.row0 = (uint32_t)(p_.row_start + lo);
    rb.n_rows = (uint16_t)(hi - lo);
    rb.flags = flags;
    rb.cbase = 0;
    rb.carry_slot = carry_slot;

    // the transposed-sum slots belong to the row-block as a whole
    {
        std::vector<const SymTile *> all_tiles;
        std::vector<const SymSeg *> all_segs;
        for (const Part &pt : parts) {
            if (pt.tiles) all_tiles.insert(all_tiles.end(), pt.tiles->begin(), pt.tiles->end());
            if (pt.symsegs) all_segs.insert(all_segs.end(), pt.symsegs->begin(), pt.symsegs->end());
        }
        if (!all_tiles.empty() || !all_segs.empty()) assign_slots(rb, &all_tiles, &all_segs);
    }
    out_.lds_doubles = std::max<uint32_t>(out_.lds_doubles, (uint32_t) rb.n_slots + rb.n_rows);
    // the bulk of the work first
    for (const Part &pt : parts)
        if (pt.tiles) emit_tile_passes(rb, *pt.tiles, (uint32_t)(pt.lo - lo));
    size_t n_sym = 0, n_unit = 0;
    {
        // row segments of the lower triangle that are read once: grouped like any others, then given
        // their slots.  With passes of their own for the long runs (split_pure_groups) those passes
        // come first, of all parts, and the mixed ones behind them: the kernel pipelines the
        // leading run of a row-block's passes
        struct SymPart { std::vector<Group> groups; std::vector<val_t> vals; uint32_t row_base; };
        std::vector<SymPart> sym_parts;
        for (const Part &pt : parts) {
            if (!pt.symsegs || pt.symsegs->empty()) continue;
            groups_.clear();
            gvals_.clear();
            for (const SymSeg *sg : *pt.symsegs) {
                add_group(sg->row - p_.row_start - pt.lo, sg->col, 1, sg->width, SPX_KIND_HORIZ, 0);
                gvals_.insert(gvals_.end(), sg->v, sg->v + sg->width);
            }
            n_sym += gvals_.size();
            if (stack_) stack_groups();
            slot_symseg_groups(rb);
            if (pure_passes_) split_pure_groups();
            sym_parts.push_back(SymPart{});
            sym_parts.back().groups.swap(groups_);
            sym_parts.back().vals.swap(gvals_);
            sym_parts.back().row_base = (uint32_t)(pt.lo - lo);
        }
        for (int round = 0; round < (pure_passes_ ? 2 : 1); ++round)
            for (SymPart &sp : sym_parts) {
                groups_.swap(sp.groups);
                gvals_.swap(sp.vals);
                emit_unit_passes(rb, true, sp.row_base,
                                 !pure_passes_ ? ALL_GROUPS : (round == 0 ? PURE_GROUPS : MIXED_GROUPS));
                groups_.swap(sp.groups);
                gvals_.swap(sp.vals);
            }
    }
    for (const Part &pt : parts) {
        groups_.clear();
        gvals_.clear();
        for (const Piece &pc : *pt.pieces) groups_from_piece(pc, pt.lo);
        if (pt.rowsegs)
            for (const RowSeg &sg : *pt.rowsegs) {
                add_group(sg.row - pt.lo, sg.col, 1, sg.width, SPX_KIND_HORIZ, 0);
                gvals_.insert(gvals_.end(), sg.v, sg.v + sg.width);
            }
        n_unit += gvals_.size();
        if (stack_) stack_groups();
        emit_unit_passes(rb, false, (uint32_t)(pt.lo - lo));
    }
    // the leftovers of all parts together (their rows count from the row-block's first)
    SingleVec merged;
    if (parts.size() > 1)
        for (const Part &pt : parts) merged.insert(merged.end(), pt.singles->begin(), pt.singles->end());
    SingleVec &singles = parts.size() > 1 ? merged : *parts.front().singles;
    const size_t n_delta = singles.size();
    emit_gather_passes(rb, singles, lo);
    // keep whole-lane over-reads of the last pass inside the arrays
    for (size_t i = 0; i < 16; ++i) out_.cidx.push_back(0);
    pad_to(out_.values, 2);
    assert(n_unit + n_delta <= 2 * SPX_MAX_RB_ELEMS * parts.size());

    out_.n_unit_elems += n_unit + n_sym;
    out_.n_delta_elems += n_delta;
    out_.nnz_stored += n_unit + n_delta + n_sym;
    out_.rbs.push_back(rb);
}

}  // namespace

// Mirror image of one unit (see gpu_emit.hpp).
static Elem mirror_unit(const Elem &e, const val_t *src, Partition &out)
{
    Elem t = e;
    t.row = e.col;
    t.col = e.row;
    switch (e.type) {
    case ENC_H: t.type = ENC_V; break;
    case ENC_V: t.type = ENC_H; break;
    case ENC_D: break;
    case ENC_AD: {
        // (r + k*d, c - k*d) mirrors to (c - k*d, r + k*d): walked from
        // its top-right end, i.e. in reverse
        const idx_t span = (idx_t)(e.size - 1) * (idx_t) e.delta;
        t.row = e.col - span;
        t.col = e.row + span;
        std::vector<val_t> rev(src, src + e.size);
        std::reverse(rev.begin(), rev.end());
        t.voff = out.pool_alloc(rev.data(), e.size);
        break;
    }
    default:
        // R x c column-major block-row  <->  c x R row-major block-col
        // (same value order); the free dimension stays in `delta`
        if (enc_is_block_row(e.type)) t.type = (uint8_t)(ENC_BC1 + (e.type - ENC_BR1));
        else t.type = (uint8_t)(ENC_BR1 + (e.type - ENC_BC1));
    }
    return t;
}

// The upper triangle as the GPU wants it: the mirrored nonzeros are cut into
// row segments (runs of consecutive columns, at most SPX_MAX_SEG_WIDTH wide)
// and equal segments in consecutive rows are stacked into dense blocks.  The
// mirror image of what CSX found in the lower triangle is mostly column
// shaped (a horizontal unit becomes a vertical one: one lane and one LDS add
// per nonzero), while eight stacked horizontal units mirror to a dense block
// that this finds again.
static void append_upper_segments(SingleVec &pts, Partition &out)
{
    sort_singles(pts);
    struct Seg { idx_t row, col; uint32_t first; uint32_t width; };
    std::vector<Seg> segs;
    for (size_t i = 0; i < pts.size();) {
        size_t j = i + 1;
        while (j < pts.size() && j - i < SPX_MAX_SEG_WIDTH && pts[j].row == pts[i].row &&
               pts[j].col == pts[j - 1].col + 1)
            ++j;
        segs.push_back(Seg{pts[i].row, pts[i].col, (uint32_t) i, (uint32_t)(j - i)});
        i = j;
    }
    sort_by_key_then(segs, [](const Seg &a) { return (int64_t) a.col; },
                     [](const Seg &a, const Seg &b) { return a.width != b.width ? a.width < b.width : a.row < b.row; });
    out.elems.reserve(out.elems.size() + segs.size());      // (an element per group of segments: at most one per segment)
    std::vector<val_t> vals;
    for (size_t i = 0; i < segs.size();) {
        const size_t w = segs[i].width;
        const size_t max_rows = 4096 / w;
        size_t j = i + 1;
        while (j < segs.size() && j - i < max_rows && segs[j].col == segs[i].col &&
               segs[j].width == w && segs[j].row == segs[j - 1].row + 1)
            ++j;
        const size_t rows = j - i;
        if (rows * w == 1) {
            const Single &s = pts[segs[i].first];
            out.elems.push_back(make_single(s.row, s.col, s.val));
        } else {
            vals.clear();
            for (size_t k = i; k < j; ++k)
                for (size_t x = 0; x < w; ++x) vals.push_back(pts[segs[k].first + x].val);
            Elem u;
            u.row = segs[i].row;
            u.col = segs[i].col;
            u.val = 0;
            u.voff = out.pool_alloc(vals.data(), vals.size());
            u.size = (uint16_t) vals.size();
            u.pad_ = 0;
            u.pad2_ = 0;
            if (rows == 1) {            // one row segment
                u.type = ENC_H;
                u.delta = 1;
            } else if (w == 1) {        // a column
                u.type = ENC_V;
                u.delta = 1;
            } else {                    // rows x w, row-major
                u.type = (uint8_t)(ENC_BC1 + (w - 1));
                u.delta = (uint32_t) rows;
            }
            out.elems.push_back(u);
        }
        i = j;
    }
}

void append_sym_expanded(const Partition &lower, Partition &out, bool remine_upper)
{
    const idx_t rs = lower.row_start;
    out.type = ENC_H;
    out.row_start = 0;
    out.nr_cols = lower.nr_cols;
    out.nr_rows = std::max<size_t>(out.nr_rows, (size_t) rs + lower.nr_rows);
    SingleVec upper;      // 1-based coordinates of the mirrored nonzeros
    if (remine_upper) upper.reserve(lower.nnz);
    for (size_t i = 0; i < lower.elems_size; ++i) {
        Elem e = lower.elems[i];
        e.row += rs;                       // global row
        if (!e.is_unit()) {
            out.elems.push_back(e);
            if (remine_upper) upper.push_back(Single{e.col, e.row, e.val});
            else out.elems.push_back(make_single(e.col, e.row, e.val));
            continue;
        }
        const val_t *src = &lower.pool[e.voff];
        e.voff = out.pool_alloc(src, e.size);
        out.elems.push_back(e);
        if (remine_upper) {
            for (size_t k = 0; k < e.size; ++k) {
                idx_t r, c;
                unit_elem_coords(e, k, r, c);
                upper.push_back(Single{c, r, src[k]});
            }
        } else {
            out.elems.push_back(mirror_unit(e, src, out));
        }
    }
    if (remine_upper) append_upper_segments(upper, out);
    out.elems_size = out.elems.size();
    out.nnz += 2 * lower.nnz;
}

// Cuts the strictly lower points into dense 8x8 tiles (eight stacked row
// segments of width 8) and the rest (in no particular order: every user sorts it).
static void extract_tiles(SingleVec &pts, std::vector<SymTile> &tiles,
                          SingleVec &rest)
{
    sort_singles(pts);
    // only full segments can be part of a tile: runs of eight consecutive columns of a row that
    // start on a column that is a multiple of eight (0-based), so that the eight column sums of a
    // tile are one aligned 64-byte piece of y.  Segments are cut as the tiles' columns would cut
    // them -- at every multiple of eight -- and the shorter ones are not looked at again.
    struct Seg { idx_t row, col; uint32_t first; };
    std::vector<Seg> full;
    for (size_t i = 0; i < pts.size();) {
        size_t j = i + 1;
        while (j < pts.size() && j - i < 8 && pts[j].row == pts[i].row &&
               pts[j].col == pts[j - 1].col + 1 && (pts[j].col - 1) % 8 != 0)
            ++j;
        if (j - i == 8) full.push_back(Seg{pts[i].row, pts[i].col, (uint32_t) i});
        i = j;
    }
    if (full.empty()) {                       // (a stencil: runs of three)
        if (rest.empty()) rest.swap(pts);
        else rest.insert(rest.end(), pts.begin(), pts.end());
        return;
    }
    sort_by_key_then(full, [](const Seg &a) { return (int64_t) a.col; },
                     [](const Seg &a, const Seg &b) { return a.row < b.row; });
    std::vector<char> in_tile(pts.size(), 0);
    size_t n_in_tiles = 0;
    for (size_t i = 0; i < full.size();) {
        size_t j = i + 1;
        while (j < full.size() && full[j].col == full[i].col && full[j].row == full[j - 1].row + 1) ++j;
        // tiles start on rows that are multiples of eight (0-based, global): a
        // row-block border then never has to cut one (see emit_gpu, step 2)
        size_t k = i;
        while (k < j && (full[k].row - 1) % 8 != 0) ++k;
        for (; k + 8 <= j; k += 8) {
            SymTile t;
            t.row0 = full[k].row - 1;          // points are 1-based
            t.col0 = full[k].col - 1;
            for (size_t r = 0; r < 8; ++r)
                for (size_t w = 0; w < 8; ++w) {
                    t.v[r * 8 + w] = pts[full[k + r].first + w].val;
                    in_tile[full[k + r].first + w] = 1;
                }
            tiles.push_back(t);
            n_in_tiles += 64;
        }
        i = j;
    }
    rest.reserve(rest.size() + pts.size() - n_in_tiles);
    for (size_t i = 0; i < pts.size(); ++i)
        if (!in_tile[i]) rest.push_back(pts[i]);
    std::sort(tiles.begin(), tiles.end(), [](const SymTile &a, const SymTile &b) {
        return a.row0 != b.row0 ? a.row0 < b.row0 : a.col0 < b.col0;
    });
}

// Joins `src` to the end of `dst` (both packed, i.e. not finalized): offsets
// of the appended row-blocks are shifted behind what `dst` already holds.
void append_stream(GpuStream &dst, GpuStream &&src, uint64_t values_placed_at)
{
    assert(!dst.pass_stride && !src.pass_stride);
    if (src.rbs.empty() && src.shared.empty()) {
        dst.lds_doubles = std::max(dst.lds_doubles, src.lds_doubles);
        return;
    }
    // (values_placed_at: the caller has copied the piece's values into dst.values at that offset
    // already -- see emit_and_upload, which does it for all pieces at once on all host threads)
    const bool placed = values_placed_at != UINT64_MAX;
    if (!placed) pad_to(dst.values, 2);
    while (dst.cidx.size() % 16) dst.cidx.push_back(0);
    const uint64_t v0 = placed ? values_placed_at : dst.values.size();
    const uint32_t p0 = (uint32_t) dst.passes.size(), d0 = (uint32_t) dst.descs.size();
    const uint32_t c0 = (uint32_t)(dst.cidx.size() / 16), s0 = (uint32_t) dst.segrows.size();
    const uint32_t k0 = dst.n_carry, sp0 = (uint32_t) dst.spill_col.size();
    if (!placed && dst.rbs.empty() && dst.values.empty()) {
        // (first piece: take the arrays as they are)
        dst.values.swap(src.values);
        dst.descs.swap(src.descs);
        dst.passes.swap(src.passes);
        dst.cidx.swap(src.cidx);
        dst.segrows.swap(src.segrows);
        dst.spill_col.swap(src.spill_col);
        dst.direct_cols.swap(src.direct_cols);
        dst.rbs.swap(src.rbs);
        dst.shared.swap(src.shared);
    } else {
        for (SpxRowBlock &rb : src.rbs) {
            rb.val_off += v0;
            rb.pass_off += p0;
            rb.desc_off += d0;
            rb.cidx_off += c0;
            rb.seg_off += s0;
            rb.spill_off += sp0;
            if (rb.flags & SPX_RB_SHARED) rb.carry_slot += k0;
        }
        for (SpxSharedRow &sr : src.shared) sr.first_slot += k0;
        if (!placed) dst.values.insert(dst.values.end(), src.values.begin(), src.values.end());
        dst.descs.insert(dst.descs.end(), src.descs.begin(), src.descs.end());
        dst.passes.insert(dst.passes.end(), src.passes.begin(), src.passes.end());
        dst.cidx.insert(dst.cidx.end(), src.cidx.begin(), src.cidx.end());
        dst.segrows.insert(dst.segrows.end(), src.segrows.begin(), src.segrows.end());
        dst.spill_col.insert(dst.spill_col.end(), src.spill_col.begin(), src.spill_col.end());
        dst.direct_cols.insert(dst.direct_cols.end(), src.direct_cols.begin(), src.direct_cols.end());
        dst.rbs.insert(dst.rbs.end(), src.rbs.begin(), src.rbs.end());
        dst.shared.insert(dst.shared.end(), src.shared.begin(), src.shared.end());
    }
    dst.n_carry += src.n_carry;
    dst.lds_doubles = std::max(dst.lds_doubles, src.lds_doubles);
    dst.nnz_stored += src.nnz_stored;
    dst.n_unit_elems += src.n_unit_elems;
    dst.n_delta_elems += src.n_delta_elems;
    dst.n_units += src.n_units;
    GpuStream().values.swap(src.values);      // release the piece's memory now
}

void build_sym_ranges(const std::vector<Partition> &lowers, const std::vector<SymRange> &ranges,
                      bool want_tiles, std::vector<Partition> &outs,
                      std::vector<std::vector<SymTile>> &tiles, unsigned nthreads,
                      std::vector<MirrorPoint> *sparse_mirror,
                      std::vector<SymSegVec> *symsegs, size_t min_run, size_t max_run)
{
    const size_t P = lowers.size(), R = ranges.size();
    auto clock = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_mark = clock(), t_step[5] = {0, 0, 0, 0, 0};
    auto lap = [&](int k) { const double t = clock(); t_step[k] = t - t_mark; t_mark = t; };
    outs.assign(R, Partition());
    tiles.assign(R, std::vector<SymTile>());
    if (!R) return;
    const idx_t nrows = ranges.back().hi;
    const size_t nr_cols = P ? lowers[0].nr_cols : (size_t) nrows;
    // range of a (0-based) row
    auto range_of = [&](idx_t row) {
        size_t lo = 0, hi = R;
        while (hi - lo > 1) {
            const size_t mid = (lo + hi) / 2;
            if (ranges[mid].lo <= row) lo = mid;
            else hi = mid;
        }
        return lo;
    };
    // the range that IS partition i (its rows), or R
    std::vector<size_t> own_range(P, R);
    for (size_t i = 0; i < P; ++i)
        for (size_t j = 0; j < R; ++j)
            if (ranges[j].lo == lowers[i].row_start) own_range[i] = j;

    // 1. every partition: its strictly lower points (1-based global), dense
    // aligned 8x8 tiles apart
    std::vector<SingleVec> rest(P);
    std::vector<std::vector<SymTile>> ptiles(P);
    parallel_for(P, nthreads, [&](size_t i) {
        const Partition &lower = lowers[i];
        const idx_t rs = lower.row_start;
        SingleVec pts;
        pts.reserve(lower.nnz);
        for (size_t k = 0; k < lower.elems_size; ++k) {
            const Elem &e = lower.elems[k];
            if (!e.is_unit()) {
                pts.push_back(Single{e.row + rs, e.col, e.val});
                continue;
            }
            const val_t *src = &lower.pool[e.voff];
            Elem g = e;
            g.row += rs;
            for (size_t q = 0; q < e.size; ++q) {
                idx_t r, c;
                unit_elem_coords(g, q, r, c);
                pts.push_back(Single{r, c, src[q]});
            }
        }
        if (want_tiles) {
            extract_tiles(pts, ptiles[i], rest[i]);
        } else {
            rest[i].swap(pts);
        }
    });

    lap(0);
    // 2. a group of eight rows that holds tiles must fit one row-block: count
    // the nonzeros of the full rows (lower + mirror image) where there are tiles
    size_t n_tiles = 0;
    for (size_t i = 0; i < P; ++i) n_tiles += ptiles[i].size();
    std::vector<char> long_row;          // rows that an over-long row-block chunking could hit
    if (n_tiles || symsegs) {
        std::vector<std::atomic<uint32_t>> cnt((size_t) nrows);
        for (auto &c : cnt) c.store(0, std::memory_order_relaxed);
        parallel_for(P, nthreads, [&](size_t i) {
            for (const Single &s : rest[i]) {
                cnt[(size_t) s.row - 1].fetch_add(1, std::memory_order_relaxed);
                cnt[(size_t) s.col - 1].fetch_add(1, std::memory_order_relaxed);
            }
            for (const SymTile &t : ptiles[i])
                for (idx_t k = 0; k < 8; ++k) {
                    cnt[(size_t)(t.row0 + k)].fetch_add(8, std::memory_order_relaxed);
                    cnt[(size_t)(t.col0 + k)].fetch_add(8, std::memory_order_relaxed);
                }
        });
        parallel_for(P, nthreads, [&](size_t i) {
            std::vector<SymTile> keep;
            for (const SymTile &t : ptiles[i]) {
                size_t need = 0;
                for (idx_t k = 0; k < 8; ++k) need += cnt[(size_t)(t.row0 + k)].load(std::memory_order_relaxed);
                if (need <= SPX_MAX_RB_ELEMS) {
                    keep.push_back(t);
                    continue;
                }
                for (idx_t r = 0; r < 8; ++r)
                    for (idx_t w = 0; w < 8; ++w)
                        rest[i].push_back(Single{t.row0 + r + 1, t.col0 + w + 1, t.v[r * 8 + w]});
            }
            ptiles[i].swap(keep);
        });
        if (symsegs) {
            long_row.assign((size_t) nrows, 0);
            for (size_t r = 0; r < (size_t) nrows; ++r)
                long_row[r] = cnt[r].load(std::memory_order_relaxed) > SPX_MAX_RB_ELEMS;
        }
    }

    lap(1);
    // 2b. read-once row segments: of what is not in a tile, the runs of three and more
    // consecutive columns of a row (cut into pieces of at most eight) are neither mirrored:
    // a lane will add their transposed products to the slots of its row-block
    std::vector<SymSegVec> psegs(P);
    if (symsegs) {
        symsegs->assign(R, SymSegVec());
        parallel_for(P, nthreads, [&](size_t i) {
            SingleVec &pts = rest[i];
            sort_singles(pts);
            SingleVec loose;
            psegs[i].reserve(pts.size() / std::max<size_t>(std::min<size_t>(min_run, 4), 1) / 2 + 1);
            for (size_t a = 0; a < pts.size();) {
                size_t b = a + 1;
                while (b < pts.size() && pts[b].row == pts[a].row && pts[b].col == pts[b - 1].col + 1) ++b;
                // a run of b - a consecutive columns: pieces of WMAX (eight, or spx.gpu.sym_segment_max), the
                // remainder if >= min_run
                // (not on rows so long that they are chunked over several row-blocks)
                size_t k = a;
                const size_t WMAX = std::min<size_t>(std::max<size_t>(max_run, min_run), SPX_MAX_SEG_WIDTH);
                const bool take = !long_row[(size_t) pts[a].row - 1];
                for (; take && b - k >= min_run; k += std::min<size_t>(WMAX, b - k)) {
                    const size_t w = std::min<size_t>(WMAX, b - k);
                    if (b - k - w > 0 && b - k - w < min_run && w == WMAX && b - k < WMAX + min_run) {
                        // (do not leave a tail shorter than that: with pieces of eight, 9 and 10 are split 5+4 / 5+5)
                        const size_t w1 = (b - k + 1) / 2;
                        SymSeg sg;
                        sg.row = pts[k].row - 1; sg.col = pts[k].col - 1; sg.width = (uint8_t) w1;
                        for (size_t q = 0; q < w1; ++q) sg.v[q] = pts[k + q].val;
                        psegs[i].push_back(sg);
                        k += w1;
                        const size_t w2 = b - k;
                        sg.row = pts[k].row - 1; sg.col = pts[k].col - 1; sg.width = (uint8_t) w2;
                        for (size_t q = 0; q < w2; ++q) sg.v[q] = pts[k + q].val;
                        psegs[i].push_back(sg);
                        k = b;
                        break;
                    }
                    SymSeg sg;
                    sg.row = pts[k].row - 1; sg.col = pts[k].col - 1; sg.width = (uint8_t) w;
                    for (size_t q = 0; q < w; ++q) sg.v[q] = pts[k + q].val;
                    psegs[i].push_back(sg);
                }
                for (; k < b; ++k) loose.push_back(pts[k]);
                a = b;
            }
            pts.swap(loose);
        });
    }

    lap(2);
    // 3. mirror image of what is not in a tile, dealt to the range of its row
    std::vector<std::vector<SingleVec>> bucket(P, std::vector<SingleVec>(R));
    parallel_for(P, nthreads, [&](size_t i) {
        size_t j = 0;
        for (const Single &s : rest[i]) {
            const idx_t row = s.col - 1;                  // 0-based row of the mirrored point
            if (!(ranges[j].lo <= row && row < ranges[j].hi)) j = range_of(row);
            bucket[i][j].push_back(Single{s.col, s.row, s.val});
        }
    });

    lap(3);
    // 4. every range: its points as row segments and blocks, rows relative to
    // the range
    std::vector<std::vector<MirrorPoint>> thin(R);
    parallel_for(R, nthreads, [&](size_t j) {
        SingleVec pts;
        size_t total = 0;
        for (size_t i = 0; i < P; ++i) total += bucket[i][j].size() + (own_range[i] == j ? rest[i].size() : 0);
        pts.reserve(total + total / 8);
        for (size_t i = 0; i < P; ++i) {
            if (own_range[i] == j) {
                pts.insert(pts.end(), rest[i].begin(), rest[i].end());
                SingleVec().swap(rest[i]);
                tiles[j].swap(ptiles[i]);
                if (symsegs) (*symsegs)[j].swap(psegs[i]);
            }
            pts.insert(pts.end(), bucket[i][j].begin(), bucket[i][j].end());
            SingleVec().swap(bucket[i][j]);
        }
        sort_singles(pts);
        // A range that holds only mirror image (rows of other processes): stretches of
        // 512 rows with fewer than 128 nonzeros are not worth a workgroup each -- they
        // leave the row-block stream for the per-row list
        bool mirror_only = true;
        for (size_t i = 0; i < P; ++i) mirror_only = mirror_only && own_range[i] != j;
        if (sparse_mirror && mirror_only && !pts.empty()) {
            SingleVec keep;
            for (size_t a = 0; a < pts.size();) {
                const idx_t chunk = (pts[a].row - 1 - ranges[j].lo) / 512;
                size_t b = a;
                while (b < pts.size() && (pts[b].row - 1 - ranges[j].lo) / 512 == chunk) ++b;
                if (b - a < 128) {
                    for (size_t k = a; k < b; ++k)
                        thin[j].push_back(MirrorPoint{pts[k].row - 1, pts[k].col - 1, pts[k].val});
                } else {
                    keep.insert(keep.end(), pts.begin() + a, pts.begin() + b);
                }
                a = b;
            }
            pts.swap(keep);
        }
        const size_t n_pts = pts.size();
        // The diagonal itself is held apart (dvalues), which splits every row's run
        // around it in two.  Where a(r,r-1) and a(r,r+1) both exist, an explicit
        // zero at (r,r) joins the runs of row r again: the triangular blocks along
        // the diagonal become dense blocks of one width instead of eight ragged ones.
        for (size_t k = 0; k + 1 < n_pts; ++k)
            if (pts[k].row == pts[k + 1].row && pts[k].col + 1 == pts[k].row &&
                pts[k + 1].col == pts[k].row + 1)
                pts.push_back(Single{pts[k].row, pts[k].row, 0.0});
        Partition &out = outs[j];
        out.type = ENC_H;
        out.row_start = ranges[j].lo;
        out.nr_rows = (size_t)(ranges[j].hi - ranges[j].lo);
        out.nr_cols = nr_cols;
        for (Single &s : pts) s.row -= ranges[j].lo;
        append_upper_segments(pts, out);
        out.elems_size = out.elems.size();
        out.nnz = n_pts;
    });
    if (sparse_mirror)
        for (size_t j = 0; j < R; ++j)       // (ranges ascend, points inside are sorted: ascending rows)
            sparse_mirror->insert(sparse_mirror->end(), thin[j].begin(), thin[j].end());
    lap(4);
    log_msg(LOG_INFO, "symmetric ranges: points and tiles %.2f s, row counts %.2f s, read-once segments %.2f s, "
            "mirror image dealt %.2f s, ranges built %.2f s\n", t_step[0], t_step[1], t_step[2], t_step[3], t_step[4]);
}

void finalize_stream(GpuStream &s, size_t nrows)
{
    if (s.pass_stride) return;
    s.n_spill = (uint32_t) s.spill_col.size();
    // tiles start on columns that are multiples of eight: the slots come in groups
    // of eight consecutive columns, one first column per group
    s.slot_group_col.clear();
    for (size_t k = 0; k < s.spill_col.size(); k += 8) {
        assert(s.spill_col[k] % 8 == 0 && k + 8 <= s.spill_col.size() && s.spill_col[k + 7] == s.spill_col[k] + 7);
        s.slot_group_col.push_back(s.spill_col[k]);
    }
    if (!s.spill_col.empty()) {
        // per row: the spill slots whose sums belong to it (counting sort by column)
        s.fix_ptr.assign(nrows + 1, 0);
        for (uint32_t c : s.spill_col) {
            if ((size_t) c >= nrows) throw FatalError("transposed-sum slot beyond the last row");
            ++s.fix_ptr[(size_t) c + 1];
        }
        for (size_t i = 0; i < nrows; ++i) s.fix_ptr[i + 1] += s.fix_ptr[i];
        s.fix_idx.resize(s.spill_col.size());
        std::vector<uint32_t> fill(s.fix_ptr.begin(), s.fix_ptr.end() - 1);
        for (size_t k = 0; k < s.spill_col.size(); ++k) s.fix_idx[fill[s.spill_col[k]]++] = (uint32_t) k;
    }
    uint32_t stride = 1;
    for (const SpxRowBlock &rb : s.rbs) stride = std::max<uint32_t>(stride, rb.n_pass);
    std::vector<SpxPass> strided(s.rbs.size() * (size_t) stride);       // (value-initialized: all zero)
    for (size_t i = 0; i < s.rbs.size(); ++i) {
        SpxRowBlock &rb = s.rbs[i];
        std::copy(s.passes.begin() + rb.pass_off, s.passes.begin() + rb.pass_off + rb.n_pass,
                  strided.begin() + i * (size_t) stride);
        rb.pass_off = (uint32_t)(i * (size_t) stride);
    }
    s.passes.swap(strided);
    s.pass_stride = stride;
}

void mark_private_rowblocks(GpuStream &s, size_t nrows, idx_t own_lo, idx_t own_hi)
{
    std::vector<char> touched(nrows + 8, 0);
    for (uint32_t g : s.slot_group_col)
        for (uint32_t c = g; c < g + 8; ++c) touched[c] = 1;
    for (const auto &d : s.direct_cols)
        for (uint32_t c = d.first; c < d.first + d.second && c < nrows; ++c) touched[c] = 1;
    for (uint32_t r : s.mirror_rows) touched[r] = 1;
    // (prefix counts: a row-block asks about a range of rows)
    std::vector<uint32_t> upto(nrows + 1, 0);
    for (size_t r = 0; r < nrows; ++r) upto[r + 1] = upto[r] + (touched[r] ? 1u : 0u);
    for (SpxRowBlock &rb : s.rbs) {
        rb.flags &= (uint8_t) ~SPX_RB_PRIVATE;
        const size_t lo = rb.row0, hi = (size_t) rb.row0 + rb.n_rows;
        if ((rb.flags & SPX_RB_SHARED) || (idx_t) lo < own_lo || (idx_t) hi > own_hi || hi > nrows) continue;
        if (upto[hi] == upto[lo]) rb.flags |= SPX_RB_PRIVATE;
    }
}

void emit_gpu(const Partition &p, const GpuEmitParams &prm, GpuStream &out, unsigned nthreads)
{
    assert(p.type == ENC_H);
    const idx_t nrows = (idx_t) p.nr_rows;
    if (nrows == 0) return;
    const size_t max_rows = std::min<size_t>(prm.max_rows, SPX_MAX_RB_ROWS);
    const size_t target = std::min<size_t>(std::max<size_t>(prm.target_elems, 64),
                                           SPX_MAX_RB_ELEMS);

    // 1. nonzeros landing in each row (units scatter below their anchor row)
    std::vector<uint32_t> cnt((size_t) nrows, 0);
    for (size_t i = 0; i < p.elems_size; ++i) {
        const Elem &e = p.elems[i];
        if (!e.is_unit()) { ++cnt[(size_t) e.row - 1]; continue; }
        for (size_t k = 0; k < e.size; ++k) {
            idx_t r, c;
            unit_elem_coords(e, k, r, c);
            assert(r >= 1 && r <= nrows);
            ++cnt[(size_t) r - 1];
        }
    }
    static const std::vector<SymTile> no_tiles;
    const std::vector<SymTile> &tiles = prm.tiles ? *prm.tiles : no_tiles;
    for (const SymTile &t : tiles)
        for (idx_t r = 0; r < 8; ++r) cnt[(size_t)(t.row0 - p.row_start + r)] += 8;
    static const SymSegVec no_segs;
    const SymSegVec &symsegs = prm.symsegs ? *prm.symsegs : no_segs;
    for (const SymSeg &sg : symsegs) cnt[(size_t)(sg.row - p.row_start)] += sg.width;

    // 2. row ranges of the row-blocks.  Symmetric tiles sit on rows that are
    // multiples of eight (global numbering); such a group of eight rows is never
    // cut: the row-block is closed in front of it when it would not fit.
    std::vector<char> tile_group((size_t) nrows / 8 + 2, 0);   // by (global row) / 8 - first group
    const idx_t g0 = p.row_start / 8;
    for (const SymTile &t : tiles) tile_group[(size_t)(t.row0 / 8 - g0)] = 1;
    auto in_group = [&](idx_t r) { return tile_group[(size_t)((p.row_start + r) / 8 - g0)] != 0; };
    std::vector<Plan> plans;
    {
        idx_t start = 0;
        size_t acc = 0;
        for (idx_t r = 0; r < nrows; ++r) {
            size_t c = cnt[(size_t) r];
            if (c > SPX_MAX_RB_ELEMS) {
                if (in_group(r)) throw FatalError("symmetric tile on an over-long row");
                if (r > start) plans.push_back(Plan{start, r, false});
                plans.push_back(Plan{r, r + 1, true});
                start = r + 1;
                acc = 0;
                continue;
            }
            const bool grouped = in_group(r);
            const bool head = grouped && (p.row_start + r) % 8 == 0;
            bool close;
            if (grouped && !head) {
                close = false;                    // inside a tile group
            } else if (head) {
                size_t need = 0;
                for (idx_t k = r; k < r + 8 && k < nrows; ++k) need += cnt[(size_t) k];
                if (need > SPX_MAX_RB_ELEMS || max_rows < 8)
                    throw FatalError("symmetric tile group exceeds a row-block");
                close = acc > 0 && (acc + need > std::max<size_t>(target, need) ||
                                    (size_t)(r - start) + 8 > max_rows);
                if (acc + need > SPX_MAX_RB_ELEMS) close = true;
            } else {
                const bool hard = acc + c > SPX_MAX_RB_ELEMS || (size_t)(r - start) >= max_rows;
                close = hard || (acc > 0 && acc + c > target);
            }
            if (close && r > start) {
                plans.push_back(Plan{start, r, false});
                start = r;
                acc = 0;
            }
            acc += c;
        }
        if (nrows > start) plans.push_back(Plan{start, nrows, false});
    }
    std::vector<uint32_t> plan_of_row((size_t) nrows);
    for (size_t i = 0; i < plans.size(); ++i)
        for (idx_t r = plans[i].row_lo; r < plans[i].row_hi; ++r)
            plan_of_row[(size_t) r] = (uint32_t) i;

    // 3. cut every unit at row-block borders
    std::vector<std::vector<Piece>> pieces(plans.size());
    std::vector<std::vector<Piece>> lin_pieces(plans.size());   // of one-nonzero-per-lane units
    std::vector<SingleVec> singles(plans.size());
    auto add_singles = [&](const Elem &u, size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; ++k) {
            idx_t r, c;
            unit_elem_coords(u, k, r, c);
            singles[plan_of_row[(size_t) r - 1]].push_back(
                Single{r - 1, c - 1, p.pool[u.voff + k]});
        }
    };
    for (size_t i = 0; i < p.elems_size; ++i) {
        const Elem &e = p.elems[i];
        if (!e.is_unit()) {
            singles[plan_of_row[(size_t) e.row - 1]].push_back(
                Single{e.row - 1, e.col - 1, e.val});
            continue;
        }
        if (enc_is_block_row(e.type)) {
            const size_t R = (size_t) enc_block_align(e.type);
            const size_t cdim = e.size / R;
            size_t ra = 0;
            while (ra < R) {
                uint32_t pl = plan_of_row[(size_t) e.row - 1 + ra];
                size_t rb = ra + 1;
                while (rb < R && plan_of_row[(size_t) e.row - 1 + rb] == pl) ++rb;
                size_t n = (rb - ra) * cdim;
                if (plans[pl].split || n <= 2) {
                    for (size_t ci = 0; ci < cdim; ++ci)
                        for (size_t j = ra; j < rb; ++j)
                            singles[pl].push_back(Single{
                                (idx_t)(e.row - 1 + j), (idx_t)(e.col - 1 + ci),
                                p.pool[e.voff + ci * R + j]});
                } else {
                    pieces[pl].push_back(Piece{(uint32_t) i, (uint16_t) ra, (uint16_t) rb});
                }
                ra = rb;
            }
            continue;
        }
        // linear and block-col units: rows are non-decreasing in k
        size_t k0 = 0;
        while (k0 < e.size) {
            idx_t r, c;
            unit_elem_coords(e, k0, r, c);
            uint32_t pl = plan_of_row[(size_t) r - 1];
            size_t k1 = k0 + 1;
            while (k1 < e.size) {
                unit_elem_coords(e, k1, r, c);
                if (plan_of_row[(size_t) r - 1] != pl) break;
                ++k1;
            }
            const bool wide_step = e.delta > SPX_MAX_STEP && !enc_is_block(e.type) &&
                                   !(e.type == ENC_H && e.delta == 1);
            const bool one_wide = !enc_is_block(e.type) && !(e.type == ENC_H && e.delta == 1);
            if (plans[pl].split || k1 - k0 <= 2 || wide_step) add_singles(e, k0, k1);
            else if (one_wide && prm.recut_linear)
                lin_pieces[pl].push_back(Piece{(uint32_t) i, (uint16_t) k0, (uint16_t) k1});
            else pieces[pl].push_back(Piece{(uint32_t) i, (uint16_t) k0, (uint16_t) k1});
            k0 = k1;
        }
    }

    // 3a. (per row-block, see recut_plan) vertical, diagonal, anti-diagonal and
    // strided units give one nonzero per lane.  Where the nonzeros of such units
    // sit next to each other along their rows (a 27-point stencil mined as
    // diagonals: six consecutive columns per row), the row-block takes them as
    // row segments instead -- the units were right for a CPU that walks one unit
    // at a time, the lanes want width.
    auto recut_plan = [&](size_t pl, std::vector<RowSeg> &rowsegs) {
        if (!prm.recut_linear) return;
        // A mined unit whose nonzeros have no neighbours along their rows gains nothing from the
        // re-cut -- each of its nonzeros would end up a leftover with a column offset and a row
        // of its own (4 bytes of index per nonzero, a gather pass) where the unit is one 8-byte
        // descriptor for the whole run: the main diagonal of a KKT system, whose rows hold their
        // stencil far from it, is such a unit.  Those stay what the miner made them.
        SingleVec pts;
        {
            struct Pt { Single s; uint32_t piece; };
            std::vector<Pt> all;
            for (size_t q = 0; q < lin_pieces[pl].size(); ++q) {
                const Piece &pc = lin_pieces[pl][q];
                const Elem &u = p.elems[pc.elem];
                for (size_t k = pc.a; k < pc.b; ++k) {
                    idx_t r, c;
                    unit_elem_coords(u, k, r, c);
                    all.push_back(Pt{Single{r - 1, c - 1, p.pool[u.voff + k]}, (uint32_t) q});
                }
            }
            sort_by_row_col(all, [](const Pt &t) { return (int64_t) t.s.row; }, [](const Pt &t) { return t.s.col; });
            std::vector<uint32_t> with_neighbour(lin_pieces[pl].size(), 0);
            for (size_t a = 0; a < all.size();) {
                size_t b = a + 1;
                while (b < all.size() && all[b].s.row == all[a].s.row && all[b].s.col == all[b - 1].s.col + 1) ++b;
                if (b - a >= 2)
                    for (size_t k = a; k < b; ++k) ++with_neighbour[all[k].piece];
                a = b;
            }
            std::vector<char> kept(lin_pieces[pl].size(), 0);
            for (size_t q = 0; q < lin_pieces[pl].size(); ++q) {
                const Piece &pc = lin_pieces[pl][q];
                // (long ones only: a short unit's nonzeros pack better as leftovers, up to eight of
                // a row per lane -- syn-cant lost 7 % with every isolated unit kept)
                if (prm.keep_units && with_neighbour[q] == 0 && pc.b - pc.a >= 32) {
                    kept[q] = 1;
                    pieces[pl].push_back(pc);
                }
            }
            pts.reserve(all.size());
            std::vector<Piece> rest;
            for (size_t q = 0; q < lin_pieces[pl].size(); ++q)
                if (!kept[q]) rest.push_back(lin_pieces[pl][q]);
            for (const Pt &t : all)
                if (!kept[t.piece]) pts.push_back(t.s);
            lin_pieces[pl].swap(rest);
        }
        size_t nseg = 0;
        for (size_t a = 0; a < pts.size(); ++nseg) {
            size_t b = a + 1;
            while (b < pts.size() && b - a < SPX_MAX_SEG_WIDTH && pts[b].row == pts[a].row &&
                   pts[b].col == pts[b - 1].col + 1)
                ++b;
            a = b;
        }
        if (pts.size() * 4 < nseg * 7) {        // below 1.75 nonzeros per segment: keep the units
            pieces[pl].insert(pieces[pl].end(), lin_pieces[pl].begin(), lin_pieces[pl].end());
            pts.clear();
        }
        // the leftover nonzeros of the row-block join in: next to a segment
        // they widen it, next to each other they form one
        if (pts.empty() && singles[pl].size() < 2) return;
        if (plans[pl].split) return;
        pts.insert(pts.end(), singles[pl].begin(), singles[pl].end());
        singles[pl].clear();
        sort_singles(pts);
        for (size_t a = 0; a < pts.size();) {
            size_t b = a + 1;
            while (b < pts.size() && b - a < SPX_MAX_SEG_WIDTH && pts[b].row == pts[a].row &&
                   pts[b].col == pts[b - 1].col + 1)
                ++b;
            if (b - a < 3) {
                // (pairs stay leftovers: a width-2 unit pass of a few lanes per
                // row-block costs more than two gathered nonzeros)
                for (size_t k = a; k < b; ++k) singles[pl].push_back(pts[k]);
            } else {
                RowSeg sg;
                sg.row = pts[a].row;
                sg.col = pts[a].col;
                sg.width = (uint8_t)(b - a);
                for (size_t k = a; k < b; ++k) sg.v[k - a] = pts[k].val;
                rowsegs.push_back(sg);
            }
            a = b;
        }
    };

    // 3b. symmetric tiles: a tile lives in the row-block of its eight rows (the
    // planner never cuts a tile group, and 8192 nonzeros per row-block bound the
    // distinct tile columns far below the slot limit)
    std::vector<std::vector<const SymTile *>> rb_tiles(plans.size());
    for (const SymTile &t : tiles) {
        const size_t r = (size_t)(t.row0 - p.row_start);
        const uint32_t pl = plan_of_row[r];
        if (plan_of_row[r + 7] != pl || plans[pl].split)
            throw FatalError("symmetric tile cut by a row-block border");
        rb_tiles[pl].push_back(&t);
    }

    // 3c. read-once row segments: to the row-block of their row (an over-long row has none:
    // build_sym_ranges leaves such rows to the mirrored path -- checked here)
    std::vector<std::vector<const SymSeg *>> rb_segs(plans.size());
    for (const SymSeg &sg : symsegs) {
        const uint32_t pl = plan_of_row[(size_t)(sg.row - p.row_start)];
        if (plans[pl].split) throw FatalError("read-once segment on an over-long row");
        rb_segs[pl].push_back(&sg);
    }

    // 4. emit: the row-blocks are independent of each other, so contiguous runs
    // of them are built by several threads into streams of their own and joined
    // in order
    // (4a. wide row-blocks: consecutive plans that hold read-once segments and no tiles go
    // side by side into one row-block of up to prm.wide_rows rows -- one y tile, one set of
    // transposed-sum slots, so that a column which several of them reach is handed to y
    // once, not once per plan)
    std::vector<std::pair<size_t, size_t>> jobs;          // plans [first, last)
    {
        // (the same for any row-block when spx.gpu.rowblock_rows allows more than 512 rows: a
        // matrix with a few nonzeros per row fills its row-blocks by rows long before it fills
        // them by nonzeros; there the planned row-blocks are joined up to the target size)
        const size_t wide_seg = std::min<size_t>(prm.wide_rows, SPX_MAX_WIDE_ROWS);
        const size_t wide_any = std::min<size_t>(prm.max_rows, SPX_MAX_WIDE_ROWS);
        auto elems_of = [&](size_t i) {
            size_t e = 0;
            for (idx_t r = plans[i].row_lo; r < plans[i].row_hi; ++r) e += cnt[(size_t) r];
            return e;
        };
        auto joinable = [&](size_t i, bool seg) {
            if (plans[i].split || !rb_tiles[i].empty()) return false;
            return seg ? wide_seg > SPX_MAX_RB_ROWS && !rb_segs[i].empty()
                       : wide_any > SPX_MAX_RB_ROWS && rb_segs[i].empty();
        };
        for (size_t i = 0; i < plans.size();) {
            size_t j = i + 1;
            if (joinable(i, true)) {
                // (... as long as the columns in front of the row-block that its segments touch
                // still fit its slots: a segment without a slot adds to y itself, one global
                // atomic per nonzero -- a 27-point stencil of edge 240 reaches 3 x (rows + 482)
                // columns, which overflowed the former 4096 slots at 1024 rows and cost 2x)
                const idx_t row0g = p.row_start + plans[i].row_lo;
                auto seg_groups = [&](size_t q, std::vector<idx_t> &g) {
                    g.clear();
                    for (const SymSeg *sg : rb_segs[q])
                        for (idx_t c = sg->col & ~(idx_t) 7; c < sg->col + sg->width && c < row0g; c += 8) g.push_back(c);
                    std::sort(g.begin(), g.end());
                    g.erase(std::unique(g.begin(), g.end()), g.end());
                };
                std::vector<idx_t> uni, g, merged;
                seg_groups(i, uni);
                while (j < plans.size() && joinable(j, true) &&
                       (size_t)(plans[j].row_hi - plans[i].row_lo) <= wide_seg) {
                    seg_groups(j, g);
                    merged.clear();
                    std::set_union(uni.begin(), uni.end(), g.begin(), g.end(), std::back_inserter(merged));
                    if (merged.size() * 8 > SPX_MAX_WIDE_SLOTS) break;
                    uni.swap(merged);
                    ++j;
                }
            } else if (joinable(i, false)) {
                size_t e = elems_of(i);
                // (spx.gpu.rowblock_elems beyond 8192 only takes effect here: planned row-blocks hold
                // at most 8192 nonzeros each -- 16-bit counters -- but several of them may go side by side)
                const size_t join_target = std::max<size_t>(target, std::min<size_t>(prm.target_elems, 4 * SPX_MAX_RB_ELEMS));
                while (j < plans.size() && joinable(j, false) &&
                       (size_t)(plans[j].row_hi - plans[i].row_lo) <= wide_any &&
                       e + elems_of(j) <= join_target + elems_of(j) / 2) {      // (overshoot by half a part at most)
                    e += elems_of(j);
                    ++j;
                }
            }
            jobs.emplace_back(i, j);
            i = j;
        }
    }
    auto emit_wide = [&](size_t first, size_t last, RbBuilder &bld) {
        std::vector<std::vector<RowSeg>> rowsegs(last - first);
        std::vector<RbBuilder::Part> parts;
        for (size_t i = first; i < last; ++i) {
            recut_plan(i, rowsegs[i - first]);
            parts.push_back(RbBuilder::Part{plans[i].row_lo, plans[i].row_hi, &pieces[i], &singles[i],
                                            &rb_tiles[i], &rowsegs[i - first], &rb_segs[i]});
        }
        bld.emit(parts, 0, 0);
    };
    auto emit_plan = [&](size_t i, RbBuilder &bld, GpuStream &dst) {
        const Plan &pl = plans[i];
        std::vector<RowSeg> rowsegs;
        recut_plan(i, rowsegs);
        if (prm.skip_empty && pieces[i].empty() && singles[i].empty() && rb_tiles[i].empty() &&
            rowsegs.empty() && rb_segs[i].empty())
            return;
        if (!pl.split) {
            bld.emit(pl.row_lo, pl.row_hi, pieces[i], singles[i], 0, 0, &rb_tiles[i], &rowsegs, &rb_segs[i]);
            return;
        }
        // an over-long row: everything is a single here; chunk it
        SingleVec &all = singles[i];
        std::sort(all.begin(), all.end(),
                  [](const Single &x, const Single &y) { return x.col < y.col; });
        const size_t chunk = SPX_MAX_RB_ELEMS / 2;
        SpxSharedRow sr;
        sr.row = (uint32_t)(p.row_start + pl.row_lo);
        sr.first_slot = dst.n_carry;
        sr.n_slots = 0;
        std::vector<Piece> none;
        for (size_t b = 0; b < all.size(); b += chunk) {
            size_t e2 = std::min(all.size(), b + chunk);
            SingleVec part(all.begin() + b, all.begin() + e2);
            bld.emit(pl.row_lo, pl.row_hi, none, part, SPX_RB_SHARED, dst.n_carry);
            ++dst.n_carry;
            ++sr.n_slots;
        }
        dst.shared.push_back(sr);
    };
    auto emit_job = [&](size_t k, RbBuilder &bld, GpuStream &dst) {
        if (jobs[k].second - jobs[k].first > 1) emit_wide(jobs[k].first, jobs[k].second, bld);
        else emit_plan(jobs[k].first, bld, dst);
    };
    if (nthreads <= 1 || jobs.size() < 64) {
        // (address space for the values up front -- the nonzeros plus an eighth of padding: pages come as they
        // are written, and a 200 MB array is not copied again each time it doubles)
        if (p.nnz * sizeof(val_t) >= ((size_t) 32 << 20)) out.values.reserve(out.values.size() + p.nnz + p.nnz / 8 + 1024);
        RbBuilder bld(p, out, prm.stack_segments, prm.x_window, prm.inline_desc, prm.sym_pure_passes);
        for (size_t k = 0; k < jobs.size(); ++k) emit_job(k, bld, out);
        return;
    }
    const size_t n_chunks = std::min<size_t>(jobs.size() / 16, (size_t) nthreads * 4);
    std::vector<GpuStream> locs(n_chunks);
    parallel_for(n_chunks, nthreads, [&](size_t c) {
        const size_t lo = jobs.size() * c / n_chunks, hi = jobs.size() * (c + 1) / n_chunks;
        RbBuilder bld(p, locs[c], prm.stack_segments, prm.x_window, prm.inline_desc, prm.sym_pure_passes);
        for (size_t k = lo; k < hi; ++k) emit_job(k, bld, locs[c]);
    });
    for (GpuStream &l : locs) append_stream(out, std::move(l));
}

}  // namespace spx
