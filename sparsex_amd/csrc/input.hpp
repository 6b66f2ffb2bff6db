// input.hpp -- matrix inputs (borrowed CSR arrays, Matrix Market files) and
// the nnz-balanced row partitioning that feeds the preprocessor.
//
// References: CSR wrapper include/sparsex/internals/Csr.hpp:56-68,346-365;
// Matrix Market reader with the SparseX header extensions
// include/sparsex/internals/Mmf.hpp:331-514, src/internals/Mmf.cpp:56-77;
// partitioning include/sparsex/internals/SparseInternal.hpp:117-152 and
// SparsePartition.hpp:508-541 (general) / :1087-1129 (symmetric).
#pragma once

#include "partition.hpp"

#include <fstream>
#include <memory>
#include <string>
#include <vector>

namespace spx {

struct Triplet { idx_t row, col; val_t val; };   // 1-based coordinates

// Row-major stream of nonzeros with one element of look-ahead.
class MatrixInput {
public:
    virtual ~MatrixInput() {}
    size_t nr_rows = 0, nr_cols = 0, nnz = 0;
    // The input may be a row slice of a larger matrix (one process per GPU, each
    // holding the rows it owns): its rows are [row_base, row_base + nr_rows) of
    // a matrix with global_rows rows.  0 / 0: the input is the whole matrix.
    size_t row_base = 0, global_rows = 0;
    virtual void rewind() = 0;
    // false at the end of the stream; does not advance
    virtual bool peek(Triplet &t) = 0;
    virtual void advance() = 0;
    // the input as CSR arrays, where it holds them (the partition builder then cuts them with all host threads
    // instead of walking element by element); may load the input and throw what loading throws
    virtual class CsrInput *as_csr() { return nullptr; }
};

class CsrInput : public MatrixInput {
public:
    CsrInput(const idx_t *rowptr, const idx_t *colind, const val_t *values,
             idx_t nr_rows, idx_t nr_cols, bool zero_based);
    void rewind() override;
    bool peek(Triplet &t) override;
    void advance() override;
    CsrInput *as_csr() override { return this; }

    const idx_t *rowptr_, *colind_;
    const val_t *values_;
    bool zero_based_;
private:
    void skip_empty();
    size_t row_ = 0;
    size_t pos_ = 0;
    // rows whose column indices are not ascending are served sorted
    std::vector<std::pair<idx_t, val_t>> sorted_row_;
    size_t sorted_for_row_ = (size_t) -1;
};

// Coordinate entries (1-based) sorted into CSR arrays (1-based, rows by ascending column; entries of one place by
// value, so that the result does not depend on the threads): a counting sort over the rows on all host threads.
// `spans`: the entries in pieces; `mirror`: an off-diagonal entry counts for (r, c) and (c, r) -- the stored
// triangle of a symmetric file; `in_order`: the entries come row-major sorted already and stay where they are.
// Throws FatalError for an entry outside the matrix or more entries than idx_t counts.
struct TripletSpan { const Triplet *first; size_t count; };
void csr_from_triplets(const std::vector<TripletSpan> &spans, size_t n_rows, size_t n_cols, bool mirror, bool in_order,
                       std::vector<idx_t> &rowptr, std::vector<idx_t> &colind, std::vector<val_t> &values, unsigned nthreads);

// CSR arrays of the library's own behind the MatrixInput interface (a reordered matrix, a file that was read)
class OwnedCsrInput : public MatrixInput {
public:
    std::vector<idx_t> rowptr, colind;      // 1-based
    std::vector<val_t> values;
    void adopt()                            // after the arrays were filled
    {
        csr_.reset(new CsrInput(rowptr.data(), colind.data(), values.data(), (idx_t) nr_rows, (idx_t) nr_cols, false));
        nnz = colind.size();
    }
    void rewind() override { csr_->rewind(); }
    bool peek(Triplet &t) override { return csr_->peek(t); }
    void advance() override { csr_->advance(); }
    CsrInput *as_csr() override { return csr_.get(); }
private:
    std::unique_ptr<CsrInput> csr_;
};

// A Matrix Market file.  The reference reads it entry by entry through an ifstream (Mmf.hpp:331-478) -- and a
// SuiteSparse file of a few hundred million entries takes minutes that way.  Here the file is mapped, cut at line
// ends into pieces that all host threads parse, and turned into CSR arrays (a counting sort by row, rows sorted by
// column), which the partition builder then cuts like a client's CSR input.  What the reference's reader accepts
// and rejects, and WHEN it rejects it, stays: a symmetric or not-row-sorted file is loaded, mirrored and sorted
// when it is opened (Mmf.hpp:445-478; nnz = what it then holds, :86-92); a file that promises row-major order
// (no banner, or the `row` keyword) is read when the tuner first asks for an element, must be sorted
// (:259-263) and must hold the entries its size line claims.
class MmfInput : public MatrixInput {
public:
    explicit MmfInput(const char *filename);   // throws FatalError
    void rewind() override;
    bool peek(Triplet &t) override;
    void advance() override;
    CsrInput *as_csr() override;

    bool symmetric = false;   // banner says symmetric (file holds one triangle)
    bool col_wise = true;     // entries not guaranteed row-major
    bool zero_based = false;
private:
    void load();              // the whole file into rowptr_ / colind_ / values_ (1-based), once

    std::string filename_;
    size_t declared_nnz_ = 0;
    size_t data_start_ = 0;             // offset of the first entry line
    bool loaded_ = false;
    std::vector<idx_t> rowptr_, colind_;
    std::vector<val_t> values_;
    std::unique_ptr<CsrInput> csr_;
};

// Splits the stream into `nr` row partitions of (roughly) equal nonzero
// count and materialises those in [first, last).  Partitions outside that
// range are walked but not stored (their boundaries still matter).
// Returns row_start/nr_rows of every partition through `bounds` (global row
// numbers when the input is a slice: the first partition starts at row_base).
struct PartBounds { idx_t row_start; idx_t nr_rows; size_t nnz; };

void build_partitions(MatrixInput &in, size_t nr, size_t first, size_t last,
                      std::vector<Partition> &parts,
                      std::vector<PartBounds> &bounds);

void build_partitions_sym(MatrixInput &in, size_t nr, size_t first, size_t last,
                          std::vector<PartitionSym> &parts,
                          std::vector<PartBounds> &bounds);

}  // namespace spx
