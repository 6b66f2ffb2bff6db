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

#include <sparsex_hip.h>

#include <vector>

namespace spx {

// A matrix held as sorted triplets (what a reordered input becomes).

// perm[old vertex] = new vertex for the graph given as CSR adjacency
// (undirected, no self loops).  Deterministic.
void rcm_order(size_t n, const std::vector<size_t> &adj_ptr, const std::vector<idx_t> &adj,
               std::vector<idx_t> &perm);

// Partition-aware order for a row-partitioned matrix (one process per GPU).  The rows are dealt
// to the processes in contiguous ranges by nonzeros (SparseInternal.hpp:131-144); in the order the
// application numbers its unknowns a range may couple with rows far away (a KKT matrix [H A^T; A D]
// keeps states and multipliers in separate blocks: a range of state rows reads a whole range of
// multipliers as x).  `order_perm` (perm[old] = position, e.g. Cuthill-McKee) says which rows belong
// together; owner_order() cuts that order into `world` ranges of equal weight and returns the
// permutation that moves every range's rows together WITHOUT changing their relative order inside
// the range -- runs of consecutive columns, diagonals and blocks of the original numbering survive.
void owner_order(const std::vector<idx_t> &order_perm, const std::vector<size_t> &weight, size_t world,
                 std::vector<idx_t> &perm);

// spx_hip_dist_reorder(): perm[old] = new for the CSR pattern of a square matrix; mode =
// SPX_DIST_REORDER_RCM (the reverse Cuthill-McKee order itself) or SPX_DIST_REORDER_RCM_OWNER
// (owner_order over it, weights = nonzeros per row).  `pattern_symmetric`: the pattern equals its
// transpose (then, zero-based, it is walked where it lies; else A + A^T is built).
void dist_reorder_csr(const idx_t *rowptr, const idx_t *colind, size_t n, bool zero_based, bool pattern_symmetric,
                      size_t world, int mode, std::vector<idx_t> &perm);

// Computes the RCM permutation of a square matrix and returns the permuted
// matrix (rows and columns renumbered, row-major sorted).  Returns nullptr --
// and leaves `perm` empty -- when no reordering is available (non-square
// matrix, or no off-diagonal nonzero), as the reference does (Rcm.hpp:276-280).
// (mode SPX_DIST_REORDER_RCM_OWNER with world > 1: owner_order over the RCM order, see above)
// (the result holds CSR arrays of its own: the partition builder cuts them like a client's CSR input)
OwnedCsrInput *reorder_rcm(MatrixInput &in, std::vector<idx_t> &perm, int mode = SPX_DIST_REORDER_RCM, size_t world = 1);

// max |row - col| over the nonzeros (diagnostics, tests)
size_t bandwidth(MatrixInput &in);

}  // namespace spx
