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
};

class CsrInput : public MatrixInput {
public:
    CsrInput(const idx_t *rowptr, const idx_t *colind, const val_t *values,
             idx_t nr_rows, idx_t nr_cols, bool zero_based);
    void rewind() override;
    bool peek(Triplet &t) override;
    void advance() override;

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

class MmfInput : public MatrixInput {
public:
    explicit MmfInput(const char *filename);   // throws FatalError
    void rewind() override;
    bool peek(Triplet &t) override;
    void advance() override;

    bool symmetric = false;   // banner says symmetric (file holds one triangle)
    bool col_wise = true;     // entries not guaranteed row-major
    bool zero_based = false;
private:
    bool read_line(std::vector<std::string> &args);
    bool next_from_file(Triplet &t);
    void load_all();

    std::ifstream in_;
    std::string filename_;
    size_t declared_nnz_ = 0;
    bool loaded_ = false;               // whole file in memory, sorted
    std::vector<Triplet> matrix_;
    size_t cursor_ = 0;
    std::streampos data_start_;
    // streaming state (header-less, sorted files)
    bool have_cur_ = false;
    Triplet cur_{};
    size_t streamed_ = 0;
    idx_t row_prev_ = 1, col_prev_ = 1;
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
