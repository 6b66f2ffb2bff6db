// partition.hpp -- internal matrix model of the CSX preprocessor.
//
// A Partition is a contiguous row range of the input, held as a vector of
// elements in a *current iteration order* (horizontal, vertical, diagonal,
// anti-diagonal, block-row R, block-col C).  An element is either a single
// nonzero or an already encoded unit (anchor coordinate + instantiation +
// values).  Semantics follow the reference's Element / SparsePartition
// (include/sparsex/internals/Element.hpp:104-420,
//  include/sparsex/internals/SparsePartition.hpp:45-250,680-839); the storage
// is flat (32-byte PODs + one value pool) instead of heap objects per nonzero.
#pragma once

#include "common.hpp"
#include "big_alloc.hpp"

#include <vector>

namespace spx {

struct Elem {
    idx_t row, col;     // 1-based coordinates in the current iteration order
    val_t val;          // value of a single element (size == 1)
    uint32_t voff;      // units: offset of the values in Partition::pool
    uint32_t delta;     // units: delta (linear) or the free dimension (blocks); 0 = single
    uint16_t size;      // number of nonzeros covered (1 for singles)
    uint8_t type;       // EncType of the unit, ENC_NONE for singles
    uint8_t pad_;
    uint32_t pad2_;     // (the struct's tail, named so that saved files do not depend on what memory held)

    // (left as it is on purpose: arrays of elements are sized first and filled by several threads
    // afterwards -- a zero fill on the sizing thread cost as much as the fill itself.  Every element
    // is made by make_single / make_unit, which set all fields.)
    Elem() {}

    bool is_unit() const { return delta != 0; }   // Element.hpp:214-219
};

static_assert(sizeof(Elem) == 32, "saved matrices hold arrays of Elem as they are");

typedef std::vector<Elem, BigAlloc<Elem>> ElemVec;   // (25 GB on the contract matrix: see big_alloc.hpp)
typedef std::vector<val_t, BigAlloc<val_t>> ValVec;  // (the interleaved values of the descriptor stream)

inline bool elem_less(const Elem &a, const Elem &b)
{
    return a.row < b.row || (a.row == b.row && a.col < b.col);
}

inline Elem make_single(idx_t r, idx_t c, val_t v)
{
    Elem e;
    e.row = r; e.col = c; e.val = v; e.voff = 0; e.delta = 0; e.size = 1;
    e.type = ENC_NONE; e.pad_ = 0; e.pad2_ = 0;
    return e;
}

// Coordinate maps between iteration orders (reference Xform.hpp:37-248).
// All coordinates are 1-based; nr_rows/nr_cols are those of the partition.
void xform_from_horiz(int to, idx_t &r, idx_t &c, idx_t nr_rows, idx_t nr_cols);
void xform_to_horiz(int from, idx_t &r, idx_t &c, idx_t nr_rows, idx_t nr_cols);
inline void xform(int from, int to, idx_t &r, idx_t &c, idx_t nr_rows,
                  idx_t nr_cols)
{
    if (from == to) return;
    if (from != ENC_H) xform_to_horiz(from, r, c, nr_rows, nr_cols);
    if (to != ENC_H) xform_from_horiz(to, r, c, nr_rows, nr_cols);
}

class Partition {
public:
    size_t nr_rows = 0, nr_cols = 0;
    size_t nnz = 0;              // nonzeros of the partition (never shrinks)
    int type = ENC_NONE;         // current iteration order
    idx_t row_start = 0;         // first row in the whole matrix (0-based)
    ElemVec elems;               // only the first elems_size entries are live
    size_t elems_size = 0;
    ElemVec scratch;             // transform()'s second buffer while the partition is being mined
    std::vector<idx_t> rowptr;   // rowptr.size()-1 == last non-empty row
    ValVec pool;                 // values of encoded units

    size_t rowptr_size() const { return rowptr.size(); }

    // (re)builds rowptr from the first `count` elements of `elems`
    // (SparsePartition::SetRowptr, SparsePartition.hpp:541-563)
    void set_rowptr(size_t count);

    // re-coordinates every element for iteration order t and sorts
    // (SparsePartition::Transform, SparsePartition.hpp:680-744)
    // (with_rowptr = false: for a sampling window that is only walked element by element -- in
    // column or diagonal order its row pointer would have a slot for every column of the matrix)
    void transform(int t, bool with_rowptr = true);

    // sampling windows: move rows [rs, rs+length) out / back
    // (GetWindow / PutWindow, SparsePartition.hpp:775-839)
    void get_window(idx_t rs, idx_t length, Partition &win);
    void put_window(Partition &win);

    // number of value slots a new unit of `size` elements takes from the pool
    uint32_t pool_alloc(const val_t *vals, size_t size)
    {
        uint32_t off = (uint32_t) pool.size();
        pool.insert(pool.end(), vals, vals + size);
        return off;
    }
};

// Lower-triangular partition of the symmetric path
// (SparsePartitionSym, SparsePartition.hpp:256-420,965-1129).
class PartitionSym {
public:
    Partition lower;              // strictly lower elements
    std::vector<val_t> diagonal;  // one entry per row of the partition
    Partition m1, m2;             // lower split at column row_start (DivideMatrix)

    void divide();                // lower -> m1 (col <= row_start) + m2 (rest)
    void merge();                 // m1 + m2 -> lower (row-wise interleave)
};

}  // namespace spx
