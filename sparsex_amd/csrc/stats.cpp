// stats.cpp -- see stats.hpp.
#include "stats.hpp"

#include <sstream>
#include <vector>

namespace spx {

void StatsCollection::append(const Instantiation &inst, const StatsData &d)
{
    // TypeStatsNode::AppendNode / InstStatsNode::AppendData,
    // Statistics.hpp:376-395, 316-334
    TypeStats &ts = types[inst.first];
    ts.inst[inst.second] += d;
    ts.total += d;
    total += d;
}

void StatsCollection::append_type(int type, const StatsCollection &other)
{
    auto it = other.types.find(type);
    if (it == other.types.end()) return;
    types[type] = it->second;
    recalc_total();
}

// Common post-pass of ManipulateStats (Statistics.hpp:585-622): instantiations
// whose data became all-zero are erased, then types whose aggregate is zero;
// the root aggregate is refreshed only when the *last visited type* asked for
// a recalculation.
namespace {
struct PostPass {
    std::vector<Instantiation> inst_to_erase;
    std::vector<int> types_to_erase;
    int last_recalc = 0;
};

void finish(StatsCollection &c, PostPass &pp)
{
    for (auto &i : pp.inst_to_erase) {
        auto t = c.types.find(i.first);
        if (t != c.types.end()) t->second.inst.erase(i.second);
    }
    for (int t : pp.types_to_erase) c.types.erase(t);
    if (pp.last_recalc) c.recalc_total();
}
}  // namespace

void StatsCollection::scale_all(double factor)
{
    PostPass pp;
    for (auto &tk : types) {
        int recalc = 0;
        for (auto &ik : tk.second.inst) {
            ik.second.scale(factor);
            ++recalc;
            if (ik.second.is_zero())
                pp.inst_to_erase.push_back(Instantiation(tk.first, ik.first));
        }
        if (recalc) tk.second.recalc();
        if (tk.second.total.is_zero()) pp.types_to_erase.push_back(tk.first);
        pp.last_recalc = recalc;
    }
    finish(*this, pp);
}

void StatsCollection::filter_coverage(size_t nnz, double min_coverage,
                                      std::set<Instantiation> &kept)
{
    PostPass pp;
    for (auto &tk : types) {
        int recalc = 0;
        for (auto &ik : tk.second.inst) {
            double coverage = (double) ik.second.nnz / (double) nnz;
            if (coverage < min_coverage) {
                ik.second = StatsData();
                ++recalc;
            } else {
                kept.insert(Instantiation(tk.first, ik.first));
            }
            if (ik.second.is_zero())
                pp.inst_to_erase.push_back(Instantiation(tk.first, ik.first));
        }
        if (recalc) tk.second.recalc();
        if (tk.second.total.is_zero()) pp.types_to_erase.push_back(tk.first);
        pp.last_recalc = recalc;
    }
    finish(*this, pp);
}

// ---- block splitting (what the reference's BlockSplitter does to the statistics of a block type,
// src/internals/Statistics.cpp:28-87; the arithmetic must match it, including which remainders drop out) ----
//
// The statistics of a block type are keyed by the blocks' FREE dimension (the other one is the type's
// alignment).  Two rules move what was counted under one dimension to smaller ones:
//   1. a block larger than a unit can hold (dimension x alignment > max_unit) is counted as blocks of the
//      largest dimension that fits;
//   2. the largest dimension that covers at least `min_coverage` of the matrix absorbs every larger one
//      (those are rare: as blocks of their own they would be filtered out, as pieces of the common size they
//      count).  Dimensions below it are left alone, and if no dimension passes, nothing is folded.
// Moving blocks of dimension `dim` to dimension `target` (redistribute) turns each into dim / target full blocks
// and one leftover of dim % target columns/rows -- which stays in the statistics only if it is still a block,
// i.e. at least two wide; a one-wide leftover drops out (its nonzeros are no longer counted as encoded).

namespace {

void redistribute(InstStats &stats, size_t align, size_t dim, size_t target, const StatsData whole)
{
    const size_t n_full = (dim / target) * whole.units;
    const size_t nnz_full = n_full * target * align;
    const size_t left = dim % target;
    stats[target] += StatsData(nnz_full, n_full, 0);
    if (left >= 2) stats[left] += StatsData(whole.nnz - nnz_full, whole.units, 0);
}

// the dimensions of `stats` above `floor_dim`, largest first
std::vector<size_t> dims_above(const InstStats &stats, size_t floor_dim)
{
    std::vector<size_t> out;
    for (auto it = stats.rbegin(); it != stats.rend() && it->first > floor_dim; ++it) out.push_back(it->first);
    return out;
}

// returns how many dimensions were moved (0: the statistics are unchanged)
int split_type(int type, InstStats &stats, size_t max_unit, size_t nnz, double min_coverage)
{
    if (!enc_is_block(type)) return 0;
    const size_t align = (size_t) enc_block_align(type);
    const size_t cap = max_unit / align;                 // largest free dimension a unit can hold
    int moved = 0;
    // (what a redistribution adds lies at or below its target, so the dimensions collected beforehand are
    // exactly the ones the rule applies to; each is read before anything is added under its own key)
    auto move_all = [&](const std::vector<size_t> &dims, size_t target) {
        for (size_t d : dims) redistribute(stats, align, d, target, stats[d]);
        for (size_t d : dims) stats.erase(d);
        moved += (int) dims.size();
    };
    // rule 1
    move_all(dims_above(stats, cap), cap);
    // rule 2: the anchor is looked for in the statistics as rule 1 left them
    auto covers = [&](const StatsData &d) { return !((double) d.nnz / (double) nnz < min_coverage); };
    auto anchor = stats.rbegin();
    while (anchor != stats.rend() && !covers(anchor->second)) ++anchor;
    if (anchor != stats.rend()) {
        const size_t into = anchor->first;
        move_all(dims_above(stats, into), into);
    }
    return moved;
}

}  // namespace

void StatsCollection::split_blocks(size_t max_unit, size_t nnz,
                                   double min_coverage)
{
    PostPass pp;
    for (auto &tk : types) {
        int recalc = 0;
        for (auto &ik : tk.second.inst) {
            if (ik.second.is_zero())
                pp.inst_to_erase.push_back(Instantiation(tk.first, ik.first));
        }
        recalc += split_type(tk.first, tk.second.inst, max_unit, nnz,
                             min_coverage) ? 1 : 0;
        if (recalc) tk.second.recalc();
        if (tk.second.total.is_zero()) pp.types_to_erase.push_back(tk.first);
        pp.last_recalc = recalc;
    }
    finish(*this, pp);
}

std::string StatsCollection::to_string() const
{
    std::ostringstream os;
    for (auto &tk : types) {
        os << enc_short_name(tk.first) << ":[nz:" << tk.second.total.nnz
           << ", p:" << tk.second.total.units << ", d:" << tk.second.total.deltas
           << "]: { ";
        for (auto &ik : tk.second.inst)
            os << ik.first << ":[nz:" << ik.second.nnz << ", p:" << ik.second.units
               << ", d:" << ik.second.deltas << "] ";
        os << "}\n";
    }
    os << "Total: [nz:" << total.nnz << ", p:" << total.units << ", d:"
       << total.deltas << "]";
    return os.str();
}

}  // namespace spx
