// stats.hpp -- substructure statistics: (type, delta) -> {nnz, units, deltas}.
//
// Restates the behaviour of the reference's StatsData / StatsCollection and
// its three manipulators (include/sparsex/internals/Statistics.hpp:36-822,
// src/internals/Statistics.cpp:28-87), including their bookkeeping quirks
// (which aggregates get recomputed when), because the selection heuristic
// reads those aggregates.
#pragma once

#include "common.hpp"

#include <map>
#include <set>
#include <string>

namespace spx {

struct StatsData {
    size_t nnz = 0;       // nonzeros that would be encoded
    size_t units = 0;     // units ("patterns") they would form
    size_t deltas = 0;    // delta units (cost heuristic only)

    StatsData() {}
    StatsData(size_t n, size_t u, size_t d = 0) : nnz(n), units(u), deltas(d) {}
    StatsData &operator+=(const StatsData &o)
    {
        nnz += o.nnz; units += o.units; deltas += o.deltas;
        return *this;
    }
    void scale(double f)   // truncating, like size_t *= double
    {
        nnz = (size_t)((double) nnz * f);
        units = (size_t)((double) units * f);
        deltas = (size_t)((double) deltas * f);
    }
    bool is_zero() const { return nnz == 0 && units == 0 && deltas == 0; }
};

typedef std::pair<int, size_t> Instantiation;   // (type, delta / free dim)
typedef std::map<size_t, StatsData> InstStats;

struct TypeStats {
    InstStats inst;
    StatsData total;
    void recalc()
    {
        total = StatsData();
        for (auto &kv : inst) total += kv.second;
    }
};

class StatsCollection {
public:
    std::map<int, TypeStats> types;
    StatsData total;

    void clear() { types.clear(); total = StatsData(); }
    void append(const Instantiation &inst, const StatsData &d);
    // copies one type's node from another collection (AppendStats(type, other))
    void append_type(int type, const StatsCollection &other);
    void recalc_total()
    {
        total = StatsData();
        for (auto &kv : types) total += kv.second.total;
    }

    // manipulators; each reproduces StatsCollection::ManipulateStats with the
    // corresponding StatsManipulator
    void scale_all(double factor);                               // StatsDataScaler
    void split_blocks(size_t max_unit, size_t nnz, double min_coverage);  // BlockSplitter
    void filter_coverage(size_t nnz, double min_coverage,
                         std::set<Instantiation> &kept);          // CoverageFilter

    std::string to_string() const;
};

}  // namespace spx
