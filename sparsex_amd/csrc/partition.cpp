// partition.cpp -- see partition.hpp.
#include "partition.hpp"

#include <algorithm>
#include <cassert>

namespace spx {

// ---- coordinate transforms (reference Xform.hpp:37-248) --------------------

static inline void xf_block_row(int R, idx_t &r, idx_t &c)
{
    idx_t nr = (r - 1) / R + 1;
    idx_t nc = (r - 1) % R + R * (c - 1) + 1;
    r = nr; c = nc;
}

static inline void rxf_block_row(int R, idx_t &r, idx_t &c)
{
    idx_t nr = R * (r - 1) + (c - 1) % R + 1;
    idx_t nc = (c - 1) / R + 1;
    r = nr; c = nc;
}

void xform_from_horiz(int to, idx_t &r, idx_t &c, idx_t nr_rows, idx_t nr_cols)
{
    switch (to) {
    case ENC_H:
        return;
    case ENC_V:
        std::swap(r, c);
        return;
    case ENC_D: {
        idx_t nr = nr_rows + c - r;
        idx_t nc = (c < r) ? c : r;
        assert(nr > 0);
        r = nr; c = nc;
        return;
    }
    case ENC_AD: {
        idx_t nr = r + c - 1;
        idx_t nc = (nr <= nr_cols) ? r : nr_cols - c + 1;
        r = nr; c = nc;
        return;
    }
    default:
        if (enc_is_block_row(to)) {
            xf_block_row(enc_block_align(to), r, c);
        } else if (enc_is_block_col(to)) {
            std::swap(r, c);
            xf_block_row(enc_block_align(to), r, c);
        } else {
            assert(false && "unhandled iteration order");
        }
    }
}

void xform_to_horiz(int from, idx_t &r, idx_t &c, idx_t nr_rows, idx_t nr_cols)
{
    switch (from) {
    case ENC_H:
        return;
    case ENC_V:
        std::swap(r, c);
        return;
    case ENC_D: {
        idx_t nr, nc;
        if (r < nr_rows) { nr = nr_rows + c - r; nc = c; }
        else             { nr = c; nc = r + c - nr_rows; }
        r = nr; c = nc;
        return;
    }
    case ENC_AD: {
        idx_t nr, nc;
        if (r <= nr_cols) { nr = c; nc = r - c + 1; }
        else              { nr = r + c - nr_cols; nc = nr_cols - c + 1; }
        r = nr; c = nc;
        return;
    }
    default:
        if (enc_is_block_row(from)) {
            rxf_block_row(enc_block_align(from), r, c);
        } else if (enc_is_block_col(from)) {
            rxf_block_row(enc_block_align(from), r, c);
            std::swap(r, c);
        } else {
            assert(false && "unhandled iteration order");
        }
    }
}

// ---- Partition ----------------------------------------------------------------

void Partition::set_rowptr(size_t count)
{
    // rowptr[i] = index of the first element of (1-based) row i+1; rows after
    // the last non-empty one are not represented (Builder::NewRow/Finalize,
    // SparsePartition.hpp:866-891).
    rowptr.clear();
    rowptr.push_back(0);
    idx_t row_prev = 1;
    for (size_t i = 0; i < count; ++i) {
        idx_t row = elems[i].row;
        if (row != row_prev) {
            assert(row > row_prev);
            rowptr.insert(rowptr.end(), (size_t)(row - row_prev), (idx_t) i);
            row_prev = row;
        }
    }
    if ((size_t) rowptr.back() != count) rowptr.push_back((idx_t) count);
}

// Elements into (row, col) order after their coordinates changed.  Every order the preprocessor
// asks for walks lines of the matrix -- rows, columns, diagonals, block rows -- and the elements
// of one line keep their relative order under most changes (a row-major list asked for its
// diagonals already holds each diagonal from top to bottom), so one stable counting pass over the
// new row numbers usually IS the sort; where it is not (block rows from row-major order, windows
// far shorter than their row range) the comparison sort finishes the job.  Coordinates are unique:
// either way the result is the order std::sort gives.
// (`counts`: the caller has counted the elements of every new row already -- counts[row + 1], rows
// 1-based -- while it changed the coordinates; `rowptr`: where to build the row pointer on the way
// through the ordered elements.  Returns whether it did build it.)
static bool sort_elems(ElemVec &elems, size_t n, idx_t max_row, ElemVec &scratch, std::vector<uint32_t> &counts,
                       std::vector<idx_t> *rowptr)
{
    auto key = [](const Elem &e) { return (uint64_t) (uint32_t) e.row << 32 | (uint32_t) e.col; };
    auto less = [&](const Elem &a, const Elem &b) { return key(a) < key(b); };
    if (n < 2) return false;
    if (n < 256 || n > 0xfffffff0ull || (uint64_t) max_row > 8 * (uint64_t) n) {
        std::sort(elems.begin(), elems.begin() + n, less);
        return false;
    }
    std::vector<uint32_t> &start = counts;                     // rows are 1-based
    if (start.empty()) {
        start.assign((size_t) max_row + 2, 0);
        for (size_t i = 0; i < n; ++i) ++start[(size_t) elems[i].row + 1];
    }
    for (size_t r = 1; r <= (size_t) max_row; ++r) start[r + 1] += start[r];
    // (the second buffer is kept by the partition -- fresh pages cost more than the pass -- and the
    // two change places afterwards; what lies behind the first n elements is not live in either)
    if (scratch.size() < elems.size()) scratch.resize(elems.size());
    Elem *out = scratch.data();
    for (size_t i = 0; i < n; ++i) out[start[(size_t) elems[i].row]++] = elems[i];
    // one more walk: are they in order, and (set_rowptr's loop) where do the rows begin
    bool in_order = true;
    if (rowptr) {
        rowptr->clear();
        rowptr->push_back(0);
    }
    idx_t row_prev = 1;
    for (size_t i = 0; i < n; ++i) {
        if (i && !(key(out[i - 1]) < key(out[i]))) {
            in_order = false;
            break;
        }
        if (rowptr && out[i].row != row_prev) {
            rowptr->insert(rowptr->end(), (size_t) (out[i].row - row_prev), (idx_t) i);
            row_prev = out[i].row;
        }
    }
    elems.swap(scratch);
    if (!in_order) {
        std::sort(elems.begin(), elems.begin() + n, less);
        return false;
    }
    if (rowptr && (size_t) rowptr->back() != n) rowptr->push_back((idx_t) n);
    return rowptr != nullptr;
}

// the largest row number an element can get in iteration order t (a bound, for sizing the counters)
static uint64_t row_bound(int t, uint64_t nr, uint64_t nc)
{
    if (t == ENC_H) return nr;
    if (t == ENC_V) return nc;
    if (t == ENC_D || t == ENC_AD) return nr + nc;
    if (enc_is_block_row(t)) return nr / (uint64_t) enc_block_align(t) + 1;
    if (enc_is_block_col(t)) return nc / (uint64_t) enc_block_align(t) + 1;
    return 0;
}

void Partition::transform(int t, bool with_rowptr)
{
    if (type == t) return;
    const idx_t nr = (idx_t) nr_rows, nc = (idx_t) nr_cols;
    const int from = type;
    const size_t n = elems_size;
    // the new rows are counted while the coordinates change, where the counting pass is going to be used
    const uint64_t bound = row_bound(t, nr_rows, nr_cols);
    std::vector<uint32_t> counts;
    bool counted = n >= 256 && n <= 0xfffffff0ull && bound > 0 && bound <= 8 * (uint64_t) n;
    if (counted) counts.assign((size_t) bound + 2, 0);
    idx_t max_row = 0;
    for (size_t i = 0; i < n; ++i) {
        xform(from, t, elems[i].row, elems[i].col, nr, nc);
        const idx_t r = elems[i].row;
        max_row = std::max(max_row, r);
        if (counted) {
            if (r >= 1 && (uint64_t) r <= bound) ++counts[(size_t) r + 1];
            else counted = false;                          // (not expected: the slow way then)
        }
    }
    if (!counted) counts.clear();
    // (the reference sorts band-by-band when both orders belong to the same row/column family,
    // SparsePartition.hpp:704-734)
    const bool have_rowptr = sort_elems(elems, n, max_row, scratch, counts, with_rowptr ? &rowptr : nullptr);
    if (!with_rowptr) {
        rowptr.clear();
        rowptr.push_back(0);
    } else if (n && !have_rowptr) {
        set_rowptr(n);
    }
    type = t;
}

void Partition::get_window(idx_t rs, idx_t length, Partition &win)
{
    win = Partition();
    // rows at the end may have lost all their anchors to units encoded in an
    // earlier round; a window starting beyond the last represented row is
    // empty (the reference reads past its rowptr array there)
    if ((size_t) rs >= rowptr.size() - 1) return;
    if ((size_t)(rs + length) > rowptr.size() - 1)
        length = (idx_t)(rowptr.size() - 1) - rs;
    idx_t es = rowptr[rs];
    idx_t ee = rowptr[rs + length];
    if (es == ee) return;
    win.elems.assign(elems.begin() + es, elems.begin() + ee);
    win.elems_size = (size_t)(ee - es);
    for (size_t i = 0; i < win.elems_size; ++i) win.elems[i].row -= rs;
    win.set_rowptr(win.elems_size);
    win.nr_rows = (size_t) length;
    win.nr_cols = nr_cols;
    win.nnz = win.elems_size;
    win.row_start = row_start + rs;
    win.type = type;
}

void Partition::put_window(Partition &win)
{
    assert(type == win.type);
    idx_t rs = win.row_start - row_start;
    idx_t es = rowptr[rs];
    if (type == ENC_H)
        for (size_t i = 0; i < win.elems_size; ++i) win.elems[i].row += rs;
    std::copy(win.elems.begin(), win.elems.begin() + win.elems_size,
              elems.begin() + es);
}

// ---- PartitionSym ---------------------------------------------------------------

static void split_rows(const Partition &src, Partition &dst, bool first_half)
{
    // DivideMatrix, SparsePartition.hpp:965-1024: an element goes to m1 when
    // its column lies left of the partition's own row range.
    dst = Partition();
    dst.type = ENC_H;
    dst.row_start = src.row_start;
    dst.nr_cols = src.nr_cols;
    dst.pool = ValVec();
    // (address space for all of them: pages come as they are written, and nothing is copied when the array grows)
    if (src.elems_size * sizeof(Elem) >= ((size_t) 32 << 20)) dst.elems.reserve(src.elems_size);
    for (size_t j = 0; j < src.elems_size; ++j) {
        const Elem &e = src.elems[j];
        bool left = e.col < src.row_start + 1;
        if (left == first_half) dst.elems.push_back(e);
    }
    dst.elems_size = dst.elems.size();
    dst.nnz = dst.elems_size;
    dst.set_rowptr(dst.elems_size);
    dst.nr_rows = dst.rowptr.size() - 1;
}

void PartitionSym::divide()
{
    split_rows(lower, m1, true);
    split_rows(lower, m2, false);
}

void PartitionSym::merge()
{
    // MergeMatrix, SparsePartition.hpp:1026-1074: per row, m1's elements
    // followed by m2's.  Unit values move into the merged pool.
    Partition out;
    out.type = ENC_H;
    out.row_start = lower.row_start;
    out.nr_cols = lower.nr_cols;
    out.nnz = lower.nnz;
    size_t nr = lower.rowptr.size() - 1;
    out.rowptr.clear();
    out.rowptr.push_back(0);
    out.rowptr.reserve(nr + 1);
    out.elems.reserve(m1.elems_size + m2.elems_size);
    out.pool.reserve(m1.pool.size() + m2.pool.size());
    auto take = [&](Partition &m, size_t i) {
        if (m.rowptr.size() - 1 <= i) return;
        for (idx_t j = m.rowptr[i]; j < m.rowptr[i + 1]; ++j) {
            Elem e = m.elems[j];
            if (e.is_unit())
                e.voff = out.pool_alloc(&m.pool[e.voff], e.size);
            out.elems.push_back(e);
        }
    };
    for (size_t i = 0; i < nr; ++i) {
        take(m1, i);
        take(m2, i);
        out.rowptr.push_back((idx_t) out.elems.size());
    }
    out.elems_size = out.elems.size();
    // Builder::Finalize appends nothing more: the last entry already equals
    // the element count.  Rows that lost all their anchors stay represented.
    // keep the partition's row count: it may exceed the last anchored row
    out.nr_rows = std::max(lower.nr_rows, out.rowptr.size() - 1);
    lower = std::move(out);
    m1 = Partition();
    m2 = Partition();
}

}  // namespace spx
