// stream_index.hpp -- host-side walk of the row-block descriptor stream
// (gpu_format.h): where in `values` does nonzero (row, col) live, and is a
// stream read from a file safe to hand to the kernels.
//
// Counterpart of the reference's random access into a tuned matrix
// (include/sparsex/internals/CsxGetSet.hpp:195-320 walks the ctl stream of a
// row and of the rows above it); here a row-block holds every nonzero of its
// rows, so one row-block (a few for an over-long row) is decoded per lookup.
#pragma once

#include "gpu_emit.hpp"

#include <string>
#include <vector>

namespace spx {

// Appends the positions in GpuStream::values of every stored copy of nonzero
// (row, col), 0-based global coordinates.  `s.values` itself is not read (the
// index arrays are enough), so `s` may be the index-only copy a matrix keeps
// on the host while its values live in HBM.
void stream_locate(const GpuStream &s, idx_t row, idx_t col, std::vector<size_t> &out);

// The same for the thin mirror image of a symmetric slice (GpuStream::mirror_*):
// positions in mirror_val of (row, col).
void stream_locate_mirror(const GpuStream &s, idx_t row, idx_t col, std::vector<size_t> &out);

// Rows in front of `below` that the stream adds to: rows of lanes of row-blocks
// that start in front of it, and the columns of symmetric tiles' spilled sums
// (s.spill_col, host side, still present).  Ascending, unique.  For a process
// that holds a slice of a symmetric matrix these are the rows of OTHER processes
// it contributes to -- the counterpart of the reference's conflict map
// (include/sparsex/internals/CsxBuild.hpp:400-451).
void stream_touched_rows(const GpuStream &s, idx_t below, std::vector<idx_t> &rows);

// Columns of x outside [own_lo, own_hi) that the products of this stream read, ascending, unique
// (padding lanes excluded).  For a process that holds the rows [own_lo, own_hi) of a row-partitioned
// matrix these are the entries of the OTHER processes' slices it needs as x -- all it needs of them:
// the halo of spx_hip_matvec_dist(..., SPX_DIST_HALO_X).  The reference has no counterpart (its
// threads share x); the closest is the column walk of its map, CsxBuild.hpp:432-451.
void stream_read_cols(const GpuStream &s, idx_t own_lo, idx_t own_hi, size_t ncols, std::vector<idx_t> &cols);

// Which parts of x a row-block reads: bit p of mask[i] is set when row-block i reads a column of
// [p * piece, (p + 1) * piece) (padding lanes included: a superset is fine); at most 64 pieces
// (piece * 64 >= ncols).  The host-vector entry point sends x in that order, the part of the product
// that needs the fewest new pieces first (device_spmv_host): the reference's threads read x where the
// client left it (src/internals/CsxKernels.cpp:35-61), here it has to cross PCIe first, and the rows
// of y that are done cross it the other way meanwhile.
void stream_rowblock_xpieces(const GpuStream &s, size_t ncols, size_t piece, std::vector<uint64_t> &mask, unsigned nthreads);

// Launch order for matrices whose rows read x in bands that recur at a fixed row distance (a
// 3-D stencil: the bands of the z-planes above and below; the distance S is N^2 rows).  Walking
// the row-blocks of an XCD's part plane by plane, a band comes round again S rows -- some 190
// row-blocks and 12 MB of values -- later, long after the XCD's 4 MB L2 has dropped it, and x is
// fetched through the fabric once per band instead of once.  Returns, for the row-blocks
// [lo, hi) of one part, the order in which to walk them: strips of `strip_rows` rows of a plane,
// the same strip of every plane in turn, then the next strip -- so that the row-blocks in flight
// at any time share their bands.  Empty when no such distance is found (most matrices).
// `stride_rows` receives S (0: none).  Pure reordering: row-blocks are independent of each other.
std::vector<uint32_t> stream_band_order(const GpuStream &s, size_t lo, size_t hi, size_t &stride_rows);

// Structural checks of a finalized stream (array-size relations, offsets of
// every row-block and pass inside their arrays, rows and columns inside the
// matrix, LDS budget).  Returns false and a reason on the first violation.
bool stream_validate(const GpuStream &s, size_t nrows, size_t ncols, size_t n_values,
                     std::string &why);

}  // namespace spx
