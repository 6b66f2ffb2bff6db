// sxplan.cpp -- see sxplan.hpp.
#include "sxplan.hpp"

#include "threads.hpp"

#include <cstring>

namespace spx {

namespace {

// the lanes of a pass all belong to one unit
inline bool single_unit(const SpxPass &ps)
{
    return (ps.flags & SPX_PASSF_INLINE) || ps.mask == 0;
}

// rewrites `ps` (a read-once pass of one unit) as an SX header; false: its geometry does not fit
bool make_sx_header(const GpuStream &s, const SpxRowBlock &rb, SpxPass &ps)
{
    const SpxUnitDesc &d = s.descs[(size_t) rb.desc_off + ps.rank0];
    const uint32_t slot0 = s.descs[(size_t) rb.desc_off + ps.rank0 + 1u].col0;
    const uint32_t bits = d.bits;
    const uint32_t kind = (bits >> 22) & 7u;
    const int step = (int) (bits >> 25);
    const int drow = kind == SPX_KIND_BLOCK ? 1 : (kind >= SPX_KIND_VERT ? step : 0);
    const int dcol = (kind == SPX_KIND_HORIZ || kind == SPX_KIND_DIAG) ? step : (kind == SPX_KIND_ADIAG ? -step : 0);
    const int64_t s0 = (int64_t) (((uint32_t) ps.seg0 - ((bits >> 9) & 8191u)) & 0xffffu);     // segment of lane 0
    const int64_t last = s0 + (int64_t) ps.nseg - 1;
    const int64_t row_l0 = (int64_t) ps.elem0 + (int64_t) (bits & 511u) + s0 * drow;
    const int64_t col_l0 = (int64_t) d.col0 + s0 * dcol;
    const int64_t col_last = (int64_t) d.col0 + last * dcol;
    if (row_l0 < 0 || row_l0 + ((int64_t) ps.nseg - 1) * drow >= (int64_t) rb.n_rows || row_l0 > 2047) return false;
    if (col_l0 < 0 || col_last < 0 || col_l0 > 0xffffffffll || drow > 127 || dcol < -127 || dcol > 127) return false;
    uint32_t slot_l0 = SPX_NO_SLOT;
    if (slot0 != SPX_NO_SLOT) {
        const int64_t sl = (int64_t) slot0 + s0 * dcol;
        const int64_t sl_last = (int64_t) slot0 + last * dcol;
        const int64_t top = (int64_t) rb.n_slots + rb.n_rows;
        if (sl < 0 || sl_last < 0 || sl + ps.width > top || sl_last + ps.width > top) return false;
        slot_l0 = (uint32_t) sl;
    }
    uint32_t w[6];
    std::memcpy(w, &ps, sizeof(w));
    w[0] = (uint32_t) col_l0;
    w[1] = (uint32_t) row_l0 | ((uint32_t) drow << 11) | ((uint32_t) (dcol + 128) << 18);
    w[3] = slot_l0;
    w[4] |= SPX_PASSF_SX << 24;
    std::memcpy(&ps, w, sizeof(w));
    return true;
}

}  // namespace

void plan_sym_pipeline(const GpuStream &s, SxPlan &plan, unsigned nthreads)
{
    static_assert(sizeof(SpxPass) == 24, "an SX header is a pass header");
    const size_t n = s.rbs.size();
    plan.passes = s.passes;
    plan.n_sx.assign(n, 0u);
    plan.n_rb_sx = 0;
    plan.sym_elems = plan.sx_elems = plan.sx_passes = plan.sym_passes = 0;
    constexpr size_t CHUNK = 256;
    const size_t n_chunks = (n + CHUNK - 1) / CHUNK;
    std::vector<uint64_t> sym_of(n_chunks, 0), sx_of(n_chunks, 0), np_of(n_chunks, 0), nsx_of(n_chunks, 0), nrb_of(n_chunks, 0);
    parallel_for(n_chunks, nthreads, [&](size_t c) {
        const size_t lo = c * CHUNK, hi = std::min(n, lo + CHUNK);
        for (size_t i = lo; i < hi; ++i) {
            const SpxRowBlock &rb = s.rbs[i];
            SpxPass *px = plan.passes.data() + rb.pass_off;
            bool head = true;
            uint32_t k = 0;
            for (uint32_t t = 0; t < rb.n_pass; ++t) {
                SpxPass &ps = px[t];
                if (ps.kind != SPX_PASS_SYMSEG) { head = false; continue; }
                const uint64_t elems = (uint64_t) ps.nseg * ps.width;
                sym_of[c] += elems;
                ++np_of[c];
                if (head && ps.nseg > 0 && ps.width <= 4 && single_unit(ps) && make_sx_header(s, rb, ps)) {
                    ++k;
                    sx_of[c] += elems;
                    ++nsx_of[c];
                } else {
                    head = false;
                }
            }
            plan.n_sx[i] = k;
            nrb_of[c] += k ? 1u : 0u;
        }
    });
    for (size_t c = 0; c < n_chunks; ++c) {
        plan.sym_elems += sym_of[c];
        plan.sx_elems += sx_of[c];
        plan.sym_passes += np_of[c];
        plan.sx_passes += nsx_of[c];
        plan.n_rb_sx += nrb_of[c];
    }
}

}  // namespace spx
