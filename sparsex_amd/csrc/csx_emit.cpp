// csx_emit.cpp -- see csx_emit.hpp.
#include "csx_emit.hpp"

#include <algorithm>
#include <cassert>
#include <map>

namespace spx {

unsigned long unit_pattern_id(const Elem &e)
{
    // CsxUtil.hpp:58-74
    if (enc_is_block(e.type))
        return e.type * PATTERN_ID_OFFSET + e.size / (unsigned) enc_block_align(e.type);
    return e.type * PATTERN_ID_OFFSET + e.delta;
}

namespace {

size_t delta_bytes_for(unsigned long v)   // Delta.hpp:35-48
{
    if (v <= 0xffu) return 1;
    if (v <= 0xffffu) return 2;
    if (v <= 0xffffffffu) return 4;
    return 8;
}

class Emitter {
public:
    Emitter(const Partition &p, bool full_colind, CsxStream &out)
        : p_(p), out_(out), full_colind_(full_colind) {}

    void run(bool symmetric);

private:
    void put_varint(unsigned long v)
    {
        for (;;) {
            uint8_t b = (uint8_t)(v & 0x7f);
            if (v < 0x80) { out_.ctl.push_back(b); break; }
            out_.ctl.push_back((uint8_t)(b | 0x80));
            v >>= 7;
        }
    }
    void put_fixed(unsigned long v, size_t nbytes)
    {
        for (size_t i = 0; i < nbytes; ++i) out_.ctl.push_back((uint8_t)(v >> (8 * i)));
    }
    void put_head(bool nr, size_t rowjmp, uint8_t slot, uint8_t size, unsigned long ucol)
    {
        uint8_t flag = slot;
        if (nr) flag |= 1u << 7;
        if (rowjmp) flag |= 1u << 6;
        out_.ctl.push_back(flag);
        out_.ctl.push_back(size);
        if (rowjmp) put_varint(rowjmp);
        if (full_colind_) put_fixed(ucol, sizeof(idx_t));
        else put_varint(ucol);
    }
    uint8_t slot_of(unsigned long pattern_id)
    {
        auto it = slots_.find(pattern_id);
        if (it != slots_.end()) return it->second;
        uint8_t s = next_slot_++;
        assert(s < CTL_PATTERNS_MAX && "too many patterns");
        slots_[pattern_id] = s;
        return s;
    }
    // new-row flag + row jump of the unit about to be written
    // (UpdateNewRow, CsxManager.hpp:615-633)
    void row_flags(bool &nr, size_t &rowjmp)
    {
        nr = false;
        rowjmp = 0;
        if (new_row_) {
            nr = true;
            new_row_ = false;
            if (empty_rows_) {
                rowjmp = empty_rows_ + 1;
                empty_rows_ = 0;
                out_.row_jumps = true;
            }
        }
    }
    void add_cols(std::vector<idx_t> &cols);
    void add_unit(const Elem &e);
    void update_span(const Elem &e);
    void do_row(idx_t begin, idx_t end, bool symmetric);

    const Partition &p_;
    CsxStream &out_;
    bool full_colind_;
    std::map<unsigned long, uint8_t> slots_;
    uint8_t next_slot_ = 0;
    bool new_row_ = false;
    size_t empty_rows_ = 0;
    idx_t last_col_ = 0;
    size_t span_ = 0;
};

void Emitter::add_cols(std::vector<idx_t> &cols)
{
    // a delta unit: first column as a jump from the current column, then
    // size-1 fixed-width deltas (AddCols, CsxManager.hpp:635-682)
    size_t n = cols.size();
    idx_t last = cols[n - 1];
    idx_t col_start = cols[0];
    idx_t prev = last_col_;
    for (size_t i = 0; i < n; ++i) {
        idx_t tmp = cols[i];
        cols[i] -= prev;
        prev = tmp;
    }
    last_col_ = last;
    idx_t mx = 0;
    for (size_t i = 1; i < n; ++i) mx = std::max(mx, cols[i]);
    size_t dbytes = delta_bytes_for((unsigned long) mx);
    unsigned long patt_id = dbytes << 3;           // CsxUtil.cpp:30-33
    bool nr; size_t rowjmp;
    row_flags(nr, rowjmp);
    // a negative jump (an element left of the previous unit's last column)
    // goes out sign-extended, as in the reference's size_t conversion
    unsigned long ucol = full_colind_ ? (unsigned long)(long)(col_start - 1)
                                      : (unsigned long)(long) cols[0];
    put_head(nr, rowjmp, slot_of(patt_id), (uint8_t) n, ucol);
    for (size_t i = 1; i < n; ++i) put_fixed((unsigned long) cols[i], dbytes);
    cols.clear();
}

void Emitter::add_unit(const Elem &e)
{
    // AddPattern, CsxManager.hpp:684-706
    bool nr; size_t rowjmp;
    row_flags(nr, rowjmp);
    unsigned long ucol = full_colind_ ? (unsigned long)(long)(e.col - 1)
                                      : (unsigned long)(long)(e.col - last_col_);
    put_head(nr, rowjmp, slot_of(unit_pattern_id(e)), (uint8_t) e.size, ucol);
    // horizontal units leave the column cursor on their last element, every
    // other unit on its anchor (GetLastCol, Element.hpp:657-666)
    last_col_ = e.col;
    if (e.type == ENC_H) last_col_ += (idx_t)((e.size - 1) * e.delta);
}

void Emitter::update_span(const Elem &e)
{
    // UpdateRowSpan, CsxManager.hpp:452-496
    size_t span = 0;
    if (e.type == ENC_V || e.type == ENC_D || e.type == ENC_AD)
        span = (size_t)(e.size - 1) * e.delta;
    else if (enc_is_block_row(e.type))
        span = (size_t)(e.type - ENC_BR1);
    else if (enc_is_block_col(e.type))
        span = (size_t) e.size / (size_t) enc_block_align(e.type) - 1;
    if (span > span_) span_ = span;
}

void Emitter::do_row(idx_t begin, idx_t end, bool symmetric)
{
    // DoRow / DoSymRow, CsxManager.hpp:504-613
    std::vector<idx_t> cols;
    span_ = 0;
    last_col_ = 1;
    idx_t j = begin;
    int passes = symmetric ? 2 : 1;
    for (int pass = 0; pass < passes; ++pass) {
        for (; j < end; ++j) {
            const Elem &e = p_.elems[j];
            if (symmetric && pass == 0 && !(e.col < p_.row_start + 1)) break;
            if (e.is_unit()) {
                update_span(e);
                if (!cols.empty()) add_cols(cols);
                add_unit(e);
                out_.values.insert(out_.values.end(), p_.pool.begin() + e.voff,
                                   p_.pool.begin() + e.voff + e.size);
                continue;
            }
            if (cols.size() == (size_t) CTL_SIZE_MAX) add_cols(cols);
            cols.push_back(e.col);
            out_.values.push_back(e.val);
        }
        if (!cols.empty()) add_cols(cols);
    }
}

void Emitter::run(bool symmetric)
{
    assert(p_.type == ENC_H);
    out_.values.clear();
    out_.values.reserve(p_.nnz);
    out_.ctl.clear();
    out_.nnz = (idx_t) p_.nnz;
    out_.nrows = (idx_t) p_.nr_rows;
    out_.ncols = (idx_t) p_.nr_cols;
    out_.row_start = p_.row_start;
    out_.row_jumps = false;
    out_.full_colind = full_colind_;
    out_.rows_info.assign(p_.nr_rows, RowInfo{0, 0, 0});
    new_row_ = false;      // the first row is not marked (CsxManager.hpp:336)
    empty_rows_ = 0;

    size_t nr = p_.rowptr.size() - 1;
    for (size_t i = 0; i < nr; ++i) {
        idx_t b = p_.rowptr[i], e = p_.rowptr[i + 1];
        RowInfo &ri = out_.rows_info[i];
        if (b == e) {
            if (!new_row_) {
                ri.rowptr = 0;
                new_row_ = true;     // leading empty row
            } else {
                ++empty_rows_;
                ri.rowptr = out_.rows_info[i - 1].rowptr;
            }
            ri.valptr = 0;
            ri.span = 0;
            continue;
        }
        ri.rowptr = i > 0 ? (idx_t) out_.ctl.size() : 0;
        ri.valptr = (idx_t) out_.values.size();
        do_row(b, e, symmetric);
        ri.span = (idx_t) span_;
        new_row_ = true;
    }
    for (size_t i = nr; i < p_.nr_rows; ++i) {
        out_.rows_info[i].valptr = 0;
        out_.rows_info[i].rowptr = i ? out_.rows_info[i - 1].rowptr : 0;
        out_.rows_info[i].span = 0;
    }
    assert(out_.values.size() == p_.nnz);
    for (int i = 0; i <= CTL_PATTERNS_MAX; ++i) out_.id_map[i] = -1;
    for (auto &kv : slots_) out_.id_map[kv.second] = (long) kv.first;
    out_.id_map[slots_.size()] = -1;
}

}  // namespace

void emit_csx(const Partition &p, bool full_colind, bool symmetric, CsxStream &out)
{
    Emitter em(p, full_colind, out);
    em.run(symmetric);
}

}  // namespace spx
