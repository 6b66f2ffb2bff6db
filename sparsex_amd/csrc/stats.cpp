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

// ---- BlockSplitter (Statistics.cpp:28-87) -------------------------------------

namespace {

// Redistributes the stats of blocks with free dimension var_dim into chunks
// of max_var_dim plus a remainder block.
void split_block_data(size_t fixed_dim, size_t var_dim, size_t max_var_dim,
                      const StatsData &data, InstStats &stats)
{
    size_t nr_chunks = var_dim / max_var_dim;
    size_t rem_dim = var_dim % max_var_dim;
    size_t max_block = max_var_dim * fixed_dim;
    size_t nr_max_blocks = nr_chunks * data.units;
    size_t rem_nnz = data.nnz - nr_max_blocks * max_block;
    stats[max_var_dim] += StatsData(nr_max_blocks * max_block, nr_max_blocks, 0);
    if (rem_dim >= 2)   // one-dimensional remainders are ignored
        stats[rem_dim] += StatsData(rem_nnz, data.units, 0);
}

// The reference walks the map with reverse iterators while inserting smaller
// keys; a reverse iterator steps to "the largest key below the current one
// at the time of the step", which is what these helpers do.
bool last_key(const InstStats &m, size_t &k)
{
    if (m.empty()) return false;
    k = m.rbegin()->first;
    return true;
}

bool prev_key(const InstStats &m, size_t &k)
{
    auto it = m.lower_bound(k);
    if (it == m.begin()) return false;
    --it;
    k = it->first;
    return true;
}

int split_type(int type, InstStats &stats, size_t max_unit, size_t nnz,
               double min_coverage)
{
    if (!enc_is_block(type)) return 0;
    size_t fixed_dim = (size_t) enc_block_align(type);
    size_t max_block_dim = max_unit / fixed_dim;
    int ret = 0;
    std::vector<size_t> erase;

    // 1. cut blocks larger than a unit can hold
    size_t k;
    bool ok = last_key(stats, k);
    while (ok && k * fixed_dim > max_unit) {
        StatsData d = stats[k];
        split_block_data(fixed_dim, k, max_block_dim, d, stats);
        erase.push_back(k);
        ++ret;
        ok = prev_key(stats, k);
    }
    for (size_t d : erase) stats.erase(d);
    erase.clear();

    // 2. fold larger, low-coverage dimensions into the largest dimension
    //    that passes the coverage threshold
    size_t ki, kj;
    bool oki = last_key(stats, ki);
    bool okj = last_key(stats, kj);
    while (oki) {
        const StatsData &di = stats[ki];
        if (!((double) di.nnz / (double) nnz < min_coverage)) {
            while (okj && kj >= ki &&
                   (double) stats[kj].nnz / (double) nnz < min_coverage) {
                StatsData dj = stats[kj];
                split_block_data(fixed_dim, kj, ki, dj, stats);
                erase.push_back(kj);
                ++ret;
                okj = prev_key(stats, kj);
            }
        }
        oki = prev_key(stats, ki);
    }
    for (size_t d : erase) stats.erase(d);
    return ret;
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
