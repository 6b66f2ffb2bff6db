// gpu_emit.cpp -- see gpu_emit.hpp and gpu_format.h.
#include "gpu_emit.hpp"

#include <algorithm>
#include <cassert>
#include <cstring>

namespace spx {

namespace {

// a part of a unit that lands in one row-block
struct Piece {
    uint32_t elem;     // index of the source unit in Partition::elems
    uint16_t a, b;     // linear / block-col units: element range [a, b)
                       // block-row units: row range [a, b) of the block
};

struct Single { idx_t row, col; val_t val; };   // 0-based row (partition), 0-based col

struct Plan {
    idx_t row_lo, row_hi;   // rows [lo, hi) of the partition
    bool split;             // one over-long row, chunked
};

void pad_to(std::vector<val_t> &v, size_t mult)
{
    while (v.size() % mult) v.push_back(0.0);
}

class RbBuilder {
public:
    RbBuilder(const Partition &p, GpuStream &out) : p_(p), out_(out) {}

    // emits one row-block for rows [lo, hi) from the given pieces/singles
    void emit(idx_t lo, idx_t hi, const std::vector<Piece> &pieces,
              std::vector<Single> &singles, uint8_t flags, uint32_t carry_slot);

private:
    void put_bits(const std::vector<uint32_t> &starts, size_t n);
    const Partition &p_;
    GpuStream &out_;
};

void RbBuilder::put_bits(const std::vector<uint32_t> &starts, size_t n)
{
    size_t passes = (n + SPX_PASS_ELEMS - 1) / SPX_PASS_ELEMS;
    size_t base = out_.bits.size();
    out_.bits.resize(base + passes * SPX_PASS_WORDS, 0u);
    std::vector<uint32_t> per_pass(passes, 0);
    for (uint32_t s : starts) {
        out_.bits[base + s / 32] |= 1u << (s % 32);
        ++per_pass[s / SPX_PASS_ELEMS];
    }
    uint32_t acc = 0;
    for (size_t p = 0; p < passes; ++p) {
        out_.pass_rank.push_back((uint16_t) acc);
        acc += per_pass[p];
    }
}

void RbBuilder::emit(idx_t lo, idx_t hi, const std::vector<Piece> &pieces,
                     std::vector<Single> &singles, uint8_t flags,
                     uint32_t carry_slot)
{
    SpxRowBlock rb;
    std::memset(&rb, 0, sizeof(rb));
    pad_to(out_.values, 4);
    rb.val_off = out_.values.size();
    rb.desc_off = (uint32_t) out_.descs.size();
    rb.bits_off = (uint32_t) out_.bits.size();
    rb.row0 = (uint32_t)(p_.row_start + lo);
    rb.n_rows = (uint16_t)(hi - lo);
    rb.flags = flags;
    rb.carry_slot = carry_slot;

    // ---- unit region --------------------------------------------------------
    std::vector<uint32_t> starts;
    size_t n_unit = 0;
    for (const Piece &pc : pieces) {
        const Elem &u = p_.elems[pc.elem];
        const val_t *src = &p_.pool[u.voff];
        SpxUnitDesc d;
        std::memset(&d, 0, sizeof(d));
        d.estart = (uint16_t) n_unit;
        starts.push_back((uint32_t) n_unit);
        size_t cnt;
        if (enc_is_block_row(u.type)) {
            // rows [a,b) of an R x cdim column-major block
            const size_t R = (size_t) enc_block_align(u.type);
            const size_t cdim = u.size / R;
            const size_t rr = pc.b - pc.a;
            // stored transposed: rows of cdim consecutive columns
            d.col0 = (uint32_t)(u.col - 1);
            d.row0 = (uint16_t)(u.row - 1 + pc.a - lo);
            d.mod = (uint8_t) cdim;
            for (size_t j = pc.a; j < pc.b; ++j)
                for (size_t i = 0; i < cdim; ++i)
                    out_.values.push_back(src[i * R + j]);
            cnt = rr * cdim;
        } else {
            idx_t r0, c0;
            unit_elem_coords(u, pc.a, r0, c0);
            d.col0 = (uint32_t)(c0 - 1);
            d.row0 = (uint16_t)(r0 - 1 - lo);
            cnt = pc.b - pc.a;
            if (enc_is_block_col(u.type)) {
                // whole rows of a rdim x C row-major block (cuts fall on rows)
                d.mod = (uint8_t) enc_block_align(u.type);
            } else {
                const int32_t dl = (int32_t) u.delta;
                d.mod = 0;
                d.dcol = (u.type == ENC_H || u.type == ENC_D) ? dl
                       : (u.type == ENC_AD) ? -dl : 0;
                d.drow = (int16_t)((u.type == ENC_H) ? 0 : dl);
            }
            out_.values.insert(out_.values.end(), src + pc.a, src + pc.b);
        }
        out_.descs.push_back(d);
        n_unit += cnt;
        ++out_.n_units;
    }
    assert(n_unit <= SPX_MAX_RB_ELEMS);
    rb.n_unit_elems = (uint16_t) n_unit;
    put_bits(starts, n_unit);
    pad_to(out_.values, 4);

    // ---- delta region ---------------------------------------------------------
    std::sort(singles.begin(), singles.end(), [](const Single &x, const Single &y) {
        return x.row < y.row || (x.row == y.row && x.col < y.col);
    });
    const size_t n_delta = singles.size();
    assert(n_delta <= SPX_MAX_RB_ELEMS);
    rb.n_delta_elems = (uint16_t) n_delta;
    rb.seg_off = (uint32_t) out_.segrows.size();
    while (out_.cidx.size() % 16) out_.cidx.push_back(0);
    rb.cidx_off = (uint32_t) out_.cidx.size();
    starts.clear();
    if (n_delta) {
        idx_t cmin = singles[0].col, cmax = singles[0].col;
        for (const Single &s : singles) {
            cmin = std::min(cmin, s.col);
            cmax = std::max(cmax, s.col);
        }
        rb.cbase = (uint32_t) cmin;
        rb.cidx_width = ((size_t)(cmax - cmin) < 65536) ? 2 : 4;
        idx_t prev_row = -1;
        for (size_t i = 0; i < n_delta; ++i) {
            const Single &s = singles[i];
            if (s.row != prev_row) {
                starts.push_back((uint32_t) i);
                out_.segrows.push_back((uint16_t)(s.row - lo));
                prev_row = s.row;
            }
            uint32_t off = (uint32_t)(s.col - cmin);
            if (rb.cidx_width == 2) {
                uint16_t o = (uint16_t) off;
                const uint8_t *b = reinterpret_cast<const uint8_t *>(&o);
                out_.cidx.insert(out_.cidx.end(), b, b + 2);
            } else {
                const uint8_t *b = reinterpret_cast<const uint8_t *>(&off);
                out_.cidx.insert(out_.cidx.end(), b, b + 4);
            }
            out_.values.push_back(s.val);
        }
    } else {
        rb.cidx_width = 2;
    }
    put_bits(starts, n_delta);
    // lanes read SPX_LANE_ELEMS offsets / values at once: keep the tail readable
    for (size_t i = 0; i < SPX_LANE_ELEMS * 4; ++i) out_.cidx.push_back(0);
    pad_to(out_.values, 4);

    out_.n_unit_elems += n_unit;
    out_.n_delta_elems += n_delta;
    out_.nnz_stored += n_unit + n_delta;
    out_.rbs.push_back(rb);
}

}  // namespace

void emit_gpu(const Partition &p, const GpuEmitParams &prm, GpuStream &out)
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

    // 2. row ranges of the row-blocks
    std::vector<Plan> plans;
    {
        idx_t start = 0;
        size_t acc = 0;
        for (idx_t r = 0; r < nrows; ++r) {
            size_t c = cnt[(size_t) r];
            if (c > SPX_MAX_RB_ELEMS) {
                if (r > start) plans.push_back(Plan{start, r, false});
                plans.push_back(Plan{r, r + 1, true});
                start = r + 1;
                acc = 0;
                continue;
            }
            if ((acc > 0 && acc + c > target) || (size_t)(r - start) >= max_rows) {
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
    std::vector<std::vector<Single>> singles(plans.size());
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
            if (plans[pl].split || k1 - k0 <= 2) add_singles(e, k0, k1);
            else pieces[pl].push_back(Piece{(uint32_t) i, (uint16_t) k0, (uint16_t) k1});
            k0 = k1;
        }
    }

    // 4. emit
    RbBuilder bld(p, out);
    for (size_t i = 0; i < plans.size(); ++i) {
        const Plan &pl = plans[i];
        if (!pl.split) {
            bld.emit(pl.row_lo, pl.row_hi, pieces[i], singles[i], 0, 0);
            continue;
        }
        // an over-long row: everything is a single here; chunk it
        std::vector<Single> &all = singles[i];
        std::sort(all.begin(), all.end(),
                  [](const Single &x, const Single &y) { return x.col < y.col; });
        const size_t chunk = SPX_MAX_RB_ELEMS / 2;
        SpxSharedRow sr;
        sr.row = (uint32_t)(p.row_start + pl.row_lo);
        sr.first_slot = out.n_carry;
        sr.n_slots = 0;
        std::vector<Piece> none;
        for (size_t b = 0; b < all.size(); b += chunk) {
            size_t e2 = std::min(all.size(), b + chunk);
            std::vector<Single> part(all.begin() + b, all.begin() + e2);
            bld.emit(pl.row_lo, pl.row_hi, none, part, SPX_RB_SHARED, out.n_carry);
            ++out.n_carry;
            ++sr.n_slots;
        }
        out.shared.push_back(sr);
    }
}

}  // namespace spx
