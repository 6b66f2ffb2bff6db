// reorder.hpp -- reverse Cuthill-McKee reordering of the input matrix
// (spx_mat_tune(input, SPX_MAT_REORDER)).
//
// The reference delegates to Boost.Graph's cuthill_mckee_ordering over the
// undirected graph of the off-diagonal pattern (include/sparsex/internals/
// Rcm.hpp:85-121 FindPerm, :248-289 ConstructGraph_CSR) and then applies the
// permutation to rows and columns (:291-317).  Boost is not a dependency
// here; this is an own Cuthill-McKee (pseudo-peripheral start per connected
// component, neighbours by ascending degree, whole order reversed).  The
// visiting order among equal-degree vertices is an implementation detail of
// either library, so the permutations need not be identical -- the contract
// (perm[old] = new, P A P^T tuned, vectors permuted with spx_vec_reorder) is.
#pragma once

#include "input.hpp"

#include <vector>

namespace spx {

// A matrix held as sorted triplets (what a reordered input becomes).
class TripletInput : public MatrixInput {
public:
    std::vector<Triplet> elems;   // 1-based, row-major sorted
    void rewind() override { cursor_ = 0; }
    bool peek(Triplet &t) override
    {
        if (cursor_ >= elems.size()) return false;
        t = elems[cursor_];
        return true;
    }
    void advance() override { ++cursor_; }
private:
    size_t cursor_ = 0;
};

// perm[old vertex] = new vertex for the graph given as CSR adjacency
// (undirected, no self loops).  Deterministic.
void rcm_order(size_t n, const std::vector<size_t> &adj_ptr, const std::vector<idx_t> &adj,
               std::vector<idx_t> &perm);

// Computes the RCM permutation of a square matrix and returns the permuted
// matrix (rows and columns renumbered, row-major sorted).  Returns nullptr --
// and leaves `perm` empty -- when no reordering is available (non-square
// matrix, or no off-diagonal nonzero), as the reference does (Rcm.hpp:276-280).
TripletInput *reorder_rcm(MatrixInput &in, std::vector<idx_t> &perm);

// max |row - col| over the nonzeros (diagnostics, tests)
size_t bandwidth(MatrixInput &in);

}  // namespace spx
