// input.cpp -- see input.hpp.
#include "input.hpp"
#include "threads.hpp"

#include <cerrno>
#include <charconv>
#include <chrono>
#include <cstring>
#include <fstream>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cassert>
#include <cstdlib>
#include <limits>
#include <sstream>

namespace spx {

// ---- CSR ------------------------------------------------------------------------

CsrInput::CsrInput(const idx_t *rowptr, const idx_t *colind, const val_t *values,
                   idx_t nrows, idx_t ncols, bool zero_based)
    : rowptr_(rowptr), colind_(colind), values_(values), zero_based_(zero_based)
{
    nr_rows = (size_t) nrows;
    nr_cols = (size_t) ncols;
    nnz = (size_t)(rowptr[nrows] - (zero_based ? 0 : 1));   // Csr.hpp:66
    rewind();
}

void CsrInput::rewind()
{
    row_ = 0;
    pos_ = 0;
    sorted_for_row_ = (size_t) -1;
    skip_empty();
}

void CsrInput::skip_empty()
{
    const idx_t base = zero_based_ ? 0 : 1;
    while (row_ < nr_rows && pos_ >= (size_t)(rowptr_[row_ + 1] - base)) ++row_;
}

bool CsrInput::peek(Triplet &t)
{
    if (pos_ >= nnz || row_ >= nr_rows) return false;
    const idx_t base = zero_based_ ? 0 : 1;
    size_t rs = (size_t)(rowptr_[row_] - base), re = (size_t)(rowptr_[row_ + 1] - base);
    if (sorted_for_row_ != row_) {
        bool ascending = true;
        for (size_t j = rs + 1; j < re; ++j)
            if (colind_[j] <= colind_[j - 1]) { ascending = false; break; }
        sorted_row_.clear();
        if (!ascending) {
            for (size_t j = rs; j < re; ++j)
                sorted_row_.push_back(std::make_pair(colind_[j], values_[j]));
            std::sort(sorted_row_.begin(), sorted_row_.end());
        }
        sorted_for_row_ = row_;
    }
    t.row = (idx_t) row_ + 1;
    if (sorted_row_.empty()) {
        t.col = colind_[pos_] + (zero_based_ ? 1 : 0);
        t.val = values_[pos_];
    } else {
        t.col = sorted_row_[pos_ - rs].first + (zero_based_ ? 1 : 0);
        t.val = sorted_row_[pos_ - rs].second;
    }
    return true;
}

void CsrInput::advance()
{
    ++pos_;
    skip_empty();
}

// ---- Matrix Market ------------------------------------------------------------------

static std::string trim(const std::string &s)
{
    size_t a = s.find_first_not_of(" \t\r\n");
    if (a == std::string::npos) return "";
    size_t b = s.find_last_not_of(" \t\r\n");
    return s.substr(a, b - a + 1);
}

static bool read_line(std::istream &in, std::vector<std::string> &args)
{
    std::string buff;
    if (!std::getline(in, buff)) return false;
    buff = trim(buff);
    args.clear();
    std::istringstream ss(buff);
    std::string tok;
    while (ss >> tok) args.push_back(tok);
    return true;
}

static bool parse3(const std::vector<std::string> &a, long &y, long &x, double &v)
{
    if (a.size() != 3) return false;
    char *e1 = nullptr, *e2 = nullptr, *e3 = nullptr;
    y = strtol(a[0].c_str(), &e1, 10);
    x = strtol(a[1].c_str(), &e2, 10);
    v = strtod(a[2].c_str(), &e3);
    return *e1 == '\0' && *e2 == '\0' && *e3 == '\0';
}

MmfInput::MmfInput(const char *filename) : filename_(filename)
{
    std::ifstream in(filename);
    if (!in.is_open()) {
        log_msg(LOG_ERR, "MMF file error\n");
        throw FatalError("cannot open MMF file");
    }
    std::vector<std::string> args;
    if (!read_line(in, args) || args.empty()) {
        log_msg(LOG_ERR, "size line error in MMF file\n");
        throw FatalError("empty MMF file");
    }
    bool headerless = false;
    if (args[0] != "%%MatrixMarket") {
        if (args[0].length() > 2 && args[0][0] == '%' && args[0][1] == '%') {
            log_msg(LOG_ERR, "invalid header line in MMF file\n");
            throw FatalError("bad banner");
        }
        // no banner: a plain "rows cols nnz" file whose entries must already
        // be sorted row-major (Mmf.hpp:372-380)
        headerless = true;
        col_wise = false;
    } else {
        if (args.size() < 5) {
            log_msg(LOG_ERR, "less arguments in header line of MMF file\n");
            throw FatalError("short banner");
        }
        for (auto &s : args)
            std::transform(s.begin(), s.end(), s.begin(), ::tolower);
        if (args[1] != "matrix") {
            log_msg(LOG_ERR, "unsupported object in header line of MMF file\n");
            throw FatalError("banner object");
        }
        if (args[2] != "coordinate") {
            log_msg(LOG_ERR, "unsupported matrix format in header line of MMF file\n");
            throw FatalError("banner format");
        }
        if (args[4] == "general") symmetric = false;
        else if (args[4] == "symmetric") symmetric = true;
        else {
            log_msg(LOG_ERR, "unsupported symmetry in header line of MMF file\n");
            throw FatalError("banner symmetry");
        }
        for (size_t i = 5; i < args.size(); ++i) {
            if (args[i] == "0-base") zero_based = true;
            else if (args[i] == "1-base") zero_based = false;
            else if (args[i] == "column") col_wise = true;
            else if (args[i] == "row") col_wise = false;
        }
    }
    // size line (after optional comment lines)
    bool skip_comments = !headerless || (!args.empty() && args[0][0] == '%');
    if (skip_comments) {
        while (in.peek() == '%')
            in.ignore(std::numeric_limits<std::streamsize>::max(), '\n');
        if (!read_line(in, args)) {
            log_msg(LOG_ERR, "size line error in MMF file\n");
            throw FatalError("size line");
        }
    }
    long r, c; double n;
    if (!parse3(args, r, c, n) || r < 0 || c < 0 || !(n >= 0 && n < 1e15) || r > (long) std::numeric_limits<idx_t>::max() ||
        c > (long) std::numeric_limits<idx_t>::max()) {
        log_msg(LOG_ERR, "bad input, less arguments in line of MMF file\n");
        throw FatalError("size line");
    }
    nr_rows = (size_t) r;
    nr_cols = (size_t) c;
    declared_nnz_ = (size_t) n;
    const std::streampos at = in.tellg();
    data_start_ = at == std::streampos(-1) ? (size_t) -1 : (size_t) at;     // (-1: the size line was the last line)
    if (symmetric || col_wise) {
        load();
        nnz = colind_.size();    // Mmf.hpp:86-92
    } else {
        nnz = declared_nnz_;
    }
}

namespace {

inline double now_seconds()
{
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

inline bool mm_blank(char ch) { return ch == ' ' || ch == '\t' || ch == '\r'; }

// strtol's grammar on [p, lim): blanks, an optional sign, digits
inline bool mm_long(const char *&p, const char *lim, long &out)
{
    while (p < lim && mm_blank(*p)) ++p;
    bool neg = false;
    if (p < lim && (*p == '+' || *p == '-')) { neg = *p == '-'; ++p; }
    if (p >= lim || *p < '0' || *p > '9') return false;
    long v = 0;
    while (p < lim && *p >= '0' && *p <= '9') {
        v = v * 10 + (*p - '0');
        if (v > ((long) 1 << 40)) return false;            // (no index is that large)
        ++p;
    }
    out = neg ? -v : v;
    return true;
}

// strtod's grammar on [p, lim): std::from_chars where it takes the whole token, strtod itself (on a copy: the
// mapping has no terminating null) for what from_chars leaves -- a leading '+', hexadecimal, out-of-range
inline bool mm_double(const char *&p, const char *lim, double &out)
{
    while (p < lim && mm_blank(*p)) ++p;
    if (p >= lim) return false;
    const std::from_chars_result r = std::from_chars(p, lim, out);
    if (r.ec == std::errc() && (r.ptr == lim || mm_blank(*r.ptr))) {
        p = r.ptr;
        return true;
    }
    char tmp[128];
    const size_t n = std::min<size_t>((size_t) (lim - p), sizeof(tmp) - 1);
    std::memcpy(tmp, p, n);
    tmp[n] = '\0';
    char *e = nullptr;
    out = strtod(tmp, &e);
    if (e == tmp) return false;
    p += e - tmp;
    return true;
}

struct MmPiece {
    const char *begin = nullptr, *end = nullptr;
    std::vector<Triplet> got;          // entries in file order (1-based after the base shift)
    bool bad = false;                  // the line behind them does not parse
    bool sorted = true;                // row-major ascending inside the piece
};

void mm_parse_piece(MmPiece &pc, bool zero_based)
{
    const char *p = pc.begin, *const end = pc.end;
    pc.got.reserve((size_t) (end - p) / 24 + 16);
    idx_t pr = 0, pcn = 0;
    while (p < end) {
        const char *eol = static_cast<const char *>(std::memchr(p, '\n', (size_t) (end - p)));
        const char *lim = eol ? eol : end;
        const char *q = p;
        long r = 0, c = 0;
        double v = 0.0;
        bool ok = mm_long(q, lim, r) && mm_long(q, lim, c) && mm_double(q, lim, v);
        while (ok && q < lim && mm_blank(*q)) ++q;
        if (!ok || q != lim) {
            pc.bad = true;
            return;
        }
        Triplet t;
        // (a coordinate beyond the index type must not wrap into the matrix: -1 fails the range check later)
        const long R = zero_based ? r + 1 : r, Cc = zero_based ? c + 1 : c;
        const long top = (long) std::numeric_limits<idx_t>::max();
        t.row = (R < 0 || R > top) ? (idx_t) -1 : (idx_t) R;
        t.col = (Cc < 0 || Cc > top) ? (idx_t) -1 : (idx_t) Cc;
        t.val = v;
        if (t.row < pr || (t.row == pr && t.col < pcn)) pc.sorted = false;
        pr = t.row;
        pcn = t.col;
        pc.got.push_back(t);
        p = eol ? eol + 1 : end;
    }
}

struct MappedFile {
    const char *data = nullptr;
    size_t size = 0;
    int fd = -1;
    explicit MappedFile(const std::string &name)
    {
        fd = open(name.c_str(), O_RDONLY);
        struct stat st;
        if (fd < 0 || fstat(fd, &st) != 0) {
            if (fd >= 0) close(fd);
            log_msg(LOG_ERR, "MMF file error\n");
            throw FatalError("cannot open MMF file");
        }
        size = (size_t) st.st_size;
        if (size) {
            void *m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) {
                close(fd);
                log_msg(LOG_ERR, "MMF file error\n");
                throw FatalError("cannot map MMF file");
            }
            (void) madvise(m, size, MADV_SEQUENTIAL);
            data = static_cast<const char *>(m);
        }
    }
    ~MappedFile()
    {
        if (data) munmap(const_cast<char *>(data), size);
        if (fd >= 0) close(fd);
    }
    MappedFile(const MappedFile &) = delete;
    MappedFile &operator=(const MappedFile &) = delete;
};

}  // namespace

void csr_from_triplets(const std::vector<TripletSpan> &spans, size_t n_rows, size_t n_cols, bool mirror, bool in_order,
                       std::vector<idx_t> &rowptr, std::vector<idx_t> &colind, std::vector<val_t> &values, unsigned T)
{
    // coordinates inside the matrix (the reference's reader does not look; a counting sort must)
    std::atomic<bool> inside(true);
    parallel_for(spans.size(), T, [&](size_t k) {
        for (size_t i = 0; i < spans[k].count; ++i) {
            const Triplet &t = spans[k].first[i];
            if (t.row < 1 || (size_t) t.row > n_rows || t.col < 1 || (size_t) t.col > n_cols ||
                (mirror && ((size_t) t.col > n_rows || (size_t) t.row > n_cols)))
                inside.store(false, std::memory_order_relaxed);
        }
    });
    if (!inside.load()) {
        log_msg(LOG_ERR, "bad input, an entry lies outside the matrix\n");
        throw FatalError("entry outside the matrix");
    }
    // rows: how many entries each holds (a mirrored off-diagonal entry counts twice), where each starts
    std::vector<uint32_t> count(n_rows + 1, 0u);
    parallel_for(spans.size(), T, [&](size_t k) {
        for (size_t i = 0; i < spans[k].count; ++i) {
            const Triplet &t = spans[k].first[i];
            __atomic_fetch_add(&count[(size_t) t.row - 1], 1u, __ATOMIC_RELAXED);
            if (mirror && t.row != t.col) __atomic_fetch_add(&count[(size_t) t.col - 1], 1u, __ATOMIC_RELAXED);
        }
    });
    size_t total = 0;
    for (size_t r = 0; r < n_rows; ++r) total += count[r];
    if (total > (size_t) std::numeric_limits<idx_t>::max() - 1) {
        log_msg(LOG_ERR, "the matrix holds more entries than the index type counts\n");
        throw FatalError("too many entries");
    }
    rowptr.assign(n_rows + 1, 1);
    for (size_t r = 0; r < n_rows; ++r) rowptr[r + 1] = rowptr[r] + (idx_t) count[r];
    colind.resize(total);
    values.resize(total);
    if (in_order) {
        // already in place: entry i is element i
        std::vector<size_t> first(spans.size() + 1, 0);
        for (size_t k = 0; k < spans.size(); ++k) first[k + 1] = first[k] + spans[k].count;
        parallel_for(spans.size(), T, [&](size_t k) {
            size_t at = first[k];
            for (size_t i = 0; i < spans[k].count; ++i, ++at) {
                colind[at] = spans[k].first[i].col;
                values[at] = spans[k].first[i].val;
            }
        });
        return;
    }
    std::vector<uint32_t> fill(n_rows, 0u);
    parallel_for(spans.size(), T, [&](size_t k) {
        for (size_t i = 0; i < spans[k].count; ++i) {
            const Triplet &t = spans[k].first[i];
            size_t at = (size_t) (rowptr[(size_t) t.row - 1] - 1) + __atomic_fetch_add(&fill[(size_t) t.row - 1], 1u, __ATOMIC_RELAXED);
            colind[at] = t.col;
            values[at] = t.val;
            if (mirror && t.row != t.col) {
                at = (size_t) (rowptr[(size_t) t.col - 1] - 1) + __atomic_fetch_add(&fill[(size_t) t.col - 1], 1u, __ATOMIC_RELAXED);
                colind[at] = t.row;
                values[at] = t.val;
            }
        }
    });
    // every row by column (entries of one place: by value, so that the result does not depend on which thread
    // came first)
    constexpr size_t ROWS = 4096;
    parallel_for((n_rows + ROWS - 1) / ROWS, T, [&](size_t c) {
        std::vector<std::pair<idx_t, val_t>> tmp;
        for (size_t r = c * ROWS; r < std::min(n_rows, (c + 1) * ROWS); ++r) {
            const size_t a = (size_t) (rowptr[r] - 1), b = (size_t) (rowptr[r + 1] - 1);
            bool ascending = true;
            for (size_t j = a + 1; j < b && ascending; ++j) ascending = colind[j] > colind[j - 1];
            if (ascending) continue;
            tmp.clear();
            for (size_t j = a; j < b; ++j) tmp.emplace_back(colind[j], values[j]);
            std::sort(tmp.begin(), tmp.end(), [](const std::pair<idx_t, val_t> &x, const std::pair<idx_t, val_t> &y) {
                if (x.first != y.first) return x.first < y.first;
                uint64_t bx, by;
                std::memcpy(&bx, &x.second, 8);
                std::memcpy(&by, &y.second, 8);
                return bx < by;
            });
            for (size_t j = a; j < b; ++j) {
                colind[j] = tmp[j - a].first;
                values[j] = tmp[j - a].second;
            }
        }
    });
}

void MmfInput::load()
{
    if (loaded_) return;
    const double t0 = now_seconds();
    const unsigned T = host_threads();
    std::vector<MmPiece> pieces;
    {
        MappedFile f(filename_);
        const size_t lo = std::min(data_start_, f.size), len = f.size - lo;
        // pieces of about 8 MB, cut behind line ends
        const size_t n_pieces = std::max<size_t>(1, std::min<size_t>(len / ((size_t) 8 << 20) + 1, (size_t) 1 << 16));
        pieces.resize(n_pieces);
        std::vector<size_t> cut(n_pieces + 1, f.size);
        cut[0] = lo;
        for (size_t k = 1; k < n_pieces; ++k) {
            size_t at = lo + len / n_pieces * k;
            if (at <= cut[k - 1]) at = cut[k - 1];
            if (at > lo && f.data[at - 1] != '\n') {
                const void *nl = at < f.size ? std::memchr(f.data + at, '\n', f.size - at) : nullptr;
                at = nl ? (size_t) (static_cast<const char *>(nl) - f.data) + 1 : f.size;
            }
            cut[k] = at;
        }
        for (size_t k = 0; k < n_pieces; ++k) {
            pieces[k].begin = f.data + cut[k];
            pieces[k].end = f.data + std::max(cut[k], cut[k + 1]);
        }
        const bool zb = zero_based;
        parallel_for(n_pieces, T, [&](size_t k) { mm_parse_piece(pieces[k], zb); });
    }
    const double t_parsed = now_seconds();
    // the entries the size line claims, in file order: a line that does not parse among them is an error, what
    // follows them is not looked at (the reference reads that many lines and stops, Mmf.hpp:445-478)
    std::vector<size_t> first(pieces.size() + 1, 0);
    size_t have = 0;
    for (size_t k = 0; k < pieces.size(); ++k) {
        first[k] = have;
        have += pieces[k].got.size();
        if (pieces[k].bad) {
            if (have < declared_nnz_) {
                log_msg(LOG_ERR, "bad input, less arguments in line of MMF file\n");
                throw FatalError("bad entry line");
            }
            for (size_t j = k + 1; j < pieces.size(); ++j) {         // (nothing behind a bad line counts)
                first[j] = have;
                pieces[j].got.clear();
            }
            break;
        }
    }
    first[pieces.size()] = have;
    if (have < declared_nnz_) {
        log_msg(LOG_ERR, "Requesting dereference, but mmf ended (cnt: %zu/%zu).\n", have, declared_nnz_);
        throw FatalError("short MMF file");
    }
    for (size_t k = 0; k < pieces.size(); ++k) {                 // keep the first declared_nnz_ entries
        const size_t keep = first[k] >= declared_nnz_ ? 0 : std::min(pieces[k].got.size(), declared_nnz_ - first[k]);
        pieces[k].got.resize(keep);
    }
    const bool streamed = !symmetric && !col_wise;
    if (streamed) {
        // entries of a file that promises row-major order must come sorted (Mmf.hpp:259-263)
        bool sorted = true;
        const Triplet *prev = nullptr;
        for (const MmPiece &pc : pieces) {
            if (pc.got.empty()) continue;
            if (!pc.sorted && pc.got.size() > 1) {
                // (`sorted` was tracked over the whole piece: look again over what was kept)
                for (size_t i = 1; i < pc.got.size() && sorted; ++i)
                    if (pc.got[i].row < pc.got[i - 1].row || (pc.got[i].row == pc.got[i - 1].row && pc.got[i].col < pc.got[i - 1].col))
                        sorted = false;
            }
            if (prev && (pc.got.front().row < prev->row || (pc.got.front().row == prev->row && pc.got.front().col < prev->col)))
                sorted = false;
            prev = &pc.got.back();
        }
        if (!sorted) {
            log_msg(LOG_ERR, "indices are not sorted in MMF file\n");
            throw FatalError("unsorted MMF file");
        }
    }
    std::vector<TripletSpan> spans;
    for (const MmPiece &pc : pieces)
        if (!pc.got.empty()) spans.push_back(TripletSpan{pc.got.data(), pc.got.size()});
    const double t_counted = now_seconds();
    csr_from_triplets(spans, nr_rows, nr_cols, symmetric, streamed, rowptr_, colind_, values_, T);
    const size_t total = colind_.size(), n_rows = nr_rows, n_cols = nr_cols;
    pieces.clear();
    csr_.reset(new CsrInput(rowptr_.data(), colind_.data(), values_.data(), (idx_t) n_rows, (idx_t) n_cols, false));
    loaded_ = true;
    const double t_end = now_seconds();
    log_msg(LOG_INFO, "MMF file: %zu entries read, %zu elements in %zu rows, %.2f s on %u threads (parsed in %.2f s, checked in "
            "%.2f s, sorted into CSR in %.2f s)\n", declared_nnz_, total, n_rows, t_end - t0, T, t_parsed - t0, t_counted - t_parsed,
            t_end - t_counted);
}

void MmfInput::rewind()
{
    load();
    csr_->rewind();
}

bool MmfInput::peek(Triplet &t)
{
    load();
    return csr_->peek(t);
}

void MmfInput::advance()
{
    load();
    csr_->advance();
}

CsrInput *MmfInput::as_csr()
{
    load();
    return csr_.get();
}

// ---- partitioning ---------------------------------------------------------------------

namespace {

// general case: SparsePartition::SetElems, SparsePartition.hpp:508-541
size_t take_partition(MatrixInput &in, idx_t row_start, size_t limit,
                      Partition *dst, idx_t &last_row)
{
    idx_t row_prev = 1;
    size_t cnt = 0;
    Triplet t;
    if (dst) dst->elems.clear();
    while (in.peek(t)) {
        idx_t row = t.row + (idx_t) in.row_base - row_start;     // 1-based inside the partition
        if (row != row_prev) {
            if (limit && cnt >= limit) break;
            row_prev = row;
        }
        if (dst) dst->elems.push_back(make_single(row, t.col, t.val));
        ++cnt;
        in.advance();
    }
    last_row = cnt ? row_prev : 0;
    return cnt;
}

// symmetric case: SparsePartitionSym::SetElems, SparsePartition.hpp:1087-1129
size_t take_partition_sym(MatrixInput &in, idx_t row_start, size_t limit,
                          PartitionSym *dst, idx_t &last_lower_row,
                          size_t &diag_cnt)
{
    idx_t row_prev = 1;
    size_t cnt = 0;
    diag_cnt = 0;
    bool any_lower = false;
    Triplet t;
    while (in.peek(t)) {
        idx_t row = t.row + (idx_t) in.row_base - row_start;
        idx_t col = t.col;
        if (row_start + row > col) {           // strictly lower
            if (row != row_prev) {
                if (limit && diag_cnt + cnt >= limit && row_prev == row - 1) break;
                row_prev = row;
            }
            if (dst) dst->lower.elems.push_back(make_single(row, col, t.val));
            ++cnt;
            any_lower = true;
        } else if (row_start + row == col) {   // diagonal
            if (dst) dst->diagonal.push_back(t.val);
            ++diag_cnt;
        }
        in.advance();
    }
    last_lower_row = any_lower ? row_prev : 0;
    return cnt;
}

// ---- the same cuts straight from CSR arrays, filled in parallel ---------------------------------
// The element-by-element walk above is what the reference does (and what any MatrixInput gets); for
// a CSR input whose rows hold ascending columns -- the common case, and 769 M virtual calls on the
// contract matrix -- the partitions are cut from the row pointers with the SAME rule and their
// elements written by all host threads.  Every bound, count and array comes out identical
// (tests/test_encoder.py, tests/test_preproc_oracle.py, tests/test_row_slices.py pin them).

struct CsrView {
    const idx_t *rowptr, *colind;
    const val_t *values;
    size_t nrows;
    idx_t base;                       // 0 or 1
    size_t at(size_t r) const { return (size_t)(rowptr[r] - base); }      // elements in front of row r
};

// rows in ascending column order and row pointers that never step back?  (else: the general walk)
bool csr_view(MatrixInput &in, CsrView &v)
{
    CsrInput *c = in.as_csr();
    if (!c || in.nr_rows == 0) return false;
    v.rowptr = c->rowptr_;
    v.colind = c->colind_;
    v.values = c->values_;
    v.nrows = in.nr_rows;
    v.base = c->zero_based_ ? 0 : 1;
    // the row pointers first, on their own: they start at the base, never step back and end at the element
    // count -- only then do they bound what is read of colind (a pointer array like [0, 5, 1000000] with ten
    // elements must end in "element count mismatch" from the general walk, not in a read far behind colind)
    if (v.rowptr[0] != v.base || v.rowptr[v.nrows] < v.base || v.at(v.nrows) != in.nnz) return false;
    std::atomic<bool> ok(true);
    const size_t CH = 1 << 16, nch = (v.nrows + CH - 1) / CH;
    parallel_for(nch, host_threads(), [&](size_t k) {
        const size_t r1 = std::min(v.nrows, (k + 1) * CH);
        for (size_t r = k * CH; r < r1; ++r)
            if (v.rowptr[r + 1] < v.rowptr[r]) { ok = false; return; }
    });
    if (!ok.load()) return false;
    parallel_for(nch, host_threads(), [&](size_t k) {
        if (!ok.load(std::memory_order_relaxed)) return;
        const size_t r1 = std::min(v.nrows, (k + 1) * CH);
        for (size_t r = k * CH; r < r1; ++r)
            for (size_t j = v.at(r) + 1; j < v.at(r + 1); ++j)
                if (v.colind[j] <= v.colind[j - 1]) { ok = false; return; }
    });
    return ok.load();
}

// last row in [lo, hi) that holds an element, or hi if none does
size_t last_nonempty(const CsrView &v, size_t lo, size_t hi)
{
    for (size_t r = hi; r > lo; --r)
        if (v.at(r) > v.at(r - 1)) return r - 1;
    return hi;
}

void finish_partition(Partition &p, size_t got, size_t nr_cols, idx_t row_start)
{
    p.elems_size = p.elems.size();
    p.nnz = got;
    p.nr_rows = p.rowptr.size() - 1;
    p.nr_cols = nr_cols;
    p.row_start = row_start;
    p.type = ENC_H;
}

bool build_partitions_csr(MatrixInput &in, size_t nr, size_t first, size_t last,
                          std::vector<Partition> &parts, std::vector<PartBounds> &bounds)
{
    CsrView v;
    if (!csr_view(in, v)) return false;
    parts.clear();
    parts.resize(last - first);
    bounds.clear();
    const size_t total = in.nnz;
    size_t cnt = 0, start = 0;                 // `start`: first input row of the partition
    struct Range { size_t start, end, got; Partition *p; };
    std::vector<Range> ranges;
    for (size_t i = 0; i < nr; ++i) {
        const size_t limit = (total - cnt) / (nr - i);
        // rows until the partition holds `limit` elements, closed behind a row that holds some
        // (take_partition: the walk stops at the first element of the next row once cnt >= limit)
        size_t end = start, got = 0;           // rows [start, end)
        const size_t left = v.at(v.nrows) - v.at(start);
        if (left) {
            size_t e;                          // last row taken
            if (limit == 0 || left < limit) {
                e = last_nonempty(v, start, v.nrows);
            } else {
                const size_t target = v.at(start) + limit;
                size_t lo = start, hi = v.nrows - 1;         // first row r with at(r + 1) >= target
                while (lo < hi) {
                    const size_t mid = (lo + hi) / 2;
                    if (v.at(mid + 1) >= target) hi = mid;
                    else lo = mid + 1;
                }
                e = lo;
            }
            end = e + 1;
            got = v.at(end) - v.at(start);
        }
        Partition *p = (i >= first && i < last) ? &parts[i - first] : nullptr;
        ranges.push_back(Range{start, end, got, p});
        PartBounds b;
        b.row_start = (idx_t)(in.row_base + start);
        b.nr_rows = (idx_t)(end - start);
        b.nnz = got;
        bounds.push_back(b);
        start = end;
        cnt += got;
    }
    if (cnt != total) {
        log_msg(LOG_ERR, "error in input matrix (matrix has less elements than "
                "claimed)\n");
        throw FatalError("element count mismatch");
    }
    // fill: pieces of <= 64 K rows of the owned partitions, on all host threads
    struct Piece { const Range *rg; size_t r0, r1; };
    std::vector<Piece> pieces;
    // (the arrays are sized -- and thereby first touched -- by several threads: value-initialising
    // 24 GB of elements of the contract matrix on one thread took longer than filling them)
    parallel_for(ranges.size(), host_threads(), [&](size_t q) {
        const Range &rg = ranges[q];
        if (!rg.p) return;
        rg.p->elems.resize(rg.got);
        rg.p->rowptr.assign(rg.end > rg.start ? rg.end - rg.start + 1 : 1, 0);
    });
    for (const Range &rg : ranges) {
        if (!rg.p) continue;
        for (size_t r = rg.start; r < rg.end; r += (1 << 16)) pieces.push_back(Piece{&rg, r, std::min(rg.end, r + (1 << 16))});
    }
    const idx_t one_based = v.base ? 0 : 1;
    parallel_for(pieces.size(), host_threads(), [&](size_t k) {
        const Piece &pc = pieces[k];
        Partition &p = *pc.rg->p;
        const size_t e0 = v.at(pc.rg->start);
        for (size_t r = pc.r0; r < pc.r1; ++r) {
            p.rowptr[r - pc.rg->start] = (idx_t)(v.at(r) - e0);
            const idx_t row = (idx_t)(r - pc.rg->start + 1);
            for (size_t j = v.at(r); j < v.at(r + 1); ++j)
                p.elems[j - e0] = make_single(row, v.colind[j] + one_based, v.values[j]);
        }
    });
    for (const Range &rg : ranges) {
        if (!rg.p) continue;
        if (rg.end > rg.start) rg.p->rowptr[rg.end - rg.start] = (idx_t) rg.got;
        finish_partition(*rg.p, rg.got, in.nr_cols, (idx_t)(in.row_base + rg.start));
    }
    return true;
}

bool build_partitions_sym_csr(MatrixInput &in, size_t nr, size_t first, size_t last,
                              std::vector<PartitionSym> &parts, std::vector<PartBounds> &bounds)
{
    CsrView v;
    if (!csr_view(in, v)) return false;
    parts.clear();
    parts.resize(last - first);
    bounds.clear();
    const idx_t one_based = v.base ? 0 : 1;
    // per row: entries strictly below the diagonal, and whether the diagonal entry is there
    std::vector<uint32_t> lower(v.nrows, 0);
    std::vector<uint8_t> diag(v.nrows, 0);
    const size_t CH = 1 << 16, nch = (v.nrows + CH - 1) / CH;
    parallel_for(nch, host_threads(), [&](size_t k) {
        const size_t r1 = std::min(v.nrows, (k + 1) * CH);
        for (size_t r = k * CH; r < r1; ++r) {
            const idx_t g = (idx_t)(in.row_base + r + 1);                   // global row, 1-based
            const idx_t *c0 = v.colind + v.at(r), *c1 = v.colind + v.at(r + 1);
            const idx_t *d = std::lower_bound(c0, c1, (idx_t)(g - one_based));   // first column >= g
            lower[r] = (uint32_t)(d - c0);
            diag[r] = (d < c1 && *d + one_based == g) ? 1 : 0;
        }
    });
    size_t total = (in.nnz + in.nr_cols) / 2;
    if (in.global_rows) {
        total = 0;
        for (size_t r = 0; r < v.nrows; ++r) total += lower[r] + diag[r];
    }
    struct Range { size_t start, end, n_lower, n_diag; idx_t nrows; PartitionSym *p; };
    std::vector<Range> ranges;
    size_t cnt = 0, start = 0;
    for (size_t i = 0; i < nr; ++i) {
        const size_t limit = (total - cnt) / (nr - i);
        // take_partition_sym row by row: a row with strictly lower entries closes the partition in front
        // of it when the count is reached AND the row before it was the last one that held such entries
        size_t n_lower = 0, n_diag = 0, r = start;
        size_t row_prev = start;               // (input row index of relative row 1)
        bool any_lower = false;
        for (; r < v.nrows; ++r) {
            if (lower[r]) {
                if (r != row_prev) {
                    if (limit && n_diag + n_lower >= limit && row_prev + 1 == r) break;
                    row_prev = r;
                }
                n_lower += lower[r];
                any_lower = true;
            }
            n_diag += diag[r];
        }
        const idx_t last_lower = any_lower ? (idx_t)(row_prev - start + 1) : 0;
        const idx_t nrows = std::max<idx_t>(last_lower, (idx_t) n_diag);
        PartitionSym *p = (i >= first && i < last) ? &parts[i - first] : nullptr;
        ranges.push_back(Range{start, r, n_lower, n_diag, nrows, p});
        PartBounds b;
        b.row_start = (idx_t)(in.row_base + start);
        b.nr_rows = nrows;
        b.nnz = n_lower + n_diag;
        bounds.push_back(b);
        cnt += n_lower + n_diag;
        // The general walk numbers the next partition's rows from start + nrows and resumes its walk at
        // row r; with a full diagonal the two coincide (every row walked delivered its diagonal).  Where
        // they do not (rows without a diagonal entry), the general walk does the job.
        if (start + (size_t) nrows != r) return false;
        start = r;
    }
    if (cnt != total) {
        log_msg(LOG_ERR, "error in input matrix (matrix has less elements than "
                "claimed)\n");
        throw FatalError("element count mismatch");
    }
    struct Piece { const Range *rg; size_t r0, r1; };
    std::vector<Piece> pieces;
    std::vector<std::vector<size_t>> lower_at(ranges.size()), diag_at(ranges.size());   // per piece start: offsets
    for (size_t q = 0; q < ranges.size(); ++q) {
        const Range &rg = ranges[q];
        if (!rg.p) continue;
        size_t lo = 0, di = 0;
        for (size_t r = rg.start; r < rg.end; r += CH) {
            const size_t r1 = std::min(rg.end, r + CH);
            pieces.push_back(Piece{&rg, r, r1});
            lower_at[q].push_back(lo);
            diag_at[q].push_back(di);
            for (size_t x = r; x < r1; ++x) {
                lo += lower[x];
                di += diag[x];
            }
        }
    }
    parallel_for(ranges.size(), host_threads(), [&](size_t q) {
        const Range &rg = ranges[q];
        if (!rg.p) return;
        rg.p->lower.elems.resize(rg.n_lower);
        rg.p->diagonal.resize(rg.n_diag);
    });
    std::vector<size_t> piece_no(pieces.size(), 0);
    {
        size_t k = 0;
        for (size_t q = 0; q < ranges.size(); ++q)
            for (size_t j = 0; j < lower_at[q].size(); ++j) piece_no[k++] = j;
    }
    parallel_for(pieces.size(), host_threads(), [&](size_t k) {
        const Piece &pc = pieces[k];
        const size_t q = (size_t)(pc.rg - ranges.data());
        Partition &lm = pc.rg->p->lower;
        size_t lo = lower_at[q][piece_no[k]], di = diag_at[q][piece_no[k]];
        for (size_t r = pc.r0; r < pc.r1; ++r) {
            const idx_t row = (idx_t)(r - pc.rg->start + 1);
            const size_t j0 = v.at(r);
            for (uint32_t t = 0; t < lower[r]; ++t)
                lm.elems[lo++] = make_single(row, v.colind[j0 + t] + one_based, v.values[j0 + t]);
            if (diag[r]) pc.rg->p->diagonal[di++] = v.values[j0 + lower[r]];
        }
    });
    for (const Range &rg : ranges) {
        if (!rg.p) continue;
        Partition &lm = rg.p->lower;
        lm.elems_size = lm.elems.size();
        lm.set_rowptr(lm.elems_size);
        lm.nnz = rg.n_lower;
        lm.nr_rows = (size_t) rg.nrows;
        lm.nr_cols = in.nr_cols;
        lm.row_start = (idx_t)(in.row_base + rg.start);
        lm.type = ENC_H;
    }
    return true;
}

}  // namespace

void build_partitions(MatrixInput &in, size_t nr, size_t first, size_t last,
                      std::vector<Partition> &parts, std::vector<PartBounds> &bounds)
{
    if (!getenv("SPX_NO_CSR_FAST_PATH") && build_partitions_csr(in, nr, first, last, parts, bounds)) return;
    in.rewind();
    parts.clear();
    parts.resize(last - first);
    bounds.clear();
    const size_t total = in.nnz;
    size_t cnt = 0;
    idx_t row_start = (idx_t) in.row_base;
    for (size_t i = 0; i < nr; ++i) {
        size_t limit = (total - cnt) / (nr - i);
        Partition *p = (i >= first && i < last) ? &parts[i - first] : nullptr;
        idx_t last_row = 0;
        size_t got = take_partition(in, row_start, limit, p, last_row);
        if (p) {
            p->elems_size = p->elems.size();
            p->set_rowptr(p->elems_size);
            p->nnz = got;
            p->nr_rows = p->rowptr.size() - 1;
            p->nr_cols = in.nr_cols;
            p->row_start = row_start;
            p->type = ENC_H;
        }
        PartBounds b;
        b.row_start = row_start;
        b.nr_rows = last_row;
        b.nnz = got;
        bounds.push_back(b);
        row_start += last_row;
        cnt += got;
    }
    if (cnt != total) {
        log_msg(LOG_ERR, "error in input matrix (matrix has less elements than "
                "claimed)\n");
        throw FatalError("element count mismatch");
    }
}

void build_partitions_sym(MatrixInput &in, size_t nr, size_t first, size_t last,
                          std::vector<PartitionSym> &parts,
                          std::vector<PartBounds> &bounds)
{
    if ((in.global_rows ? in.global_rows : in.nr_rows) != in.nr_cols) {
        log_msg(LOG_ERR, "symmetric format requested for a non-square matrix\n");
        throw FatalError("non-square symmetric");
    }
    if (!getenv("SPX_NO_CSR_FAST_PATH") && build_partitions_sym_csr(in, nr, first, last, parts, bounds)) return;
    in.rewind();
    parts.clear();
    parts.resize(last - first);
    bounds.clear();
    // lower triangle + diagonal of a matrix given in full; presumes a full
    // diagonal (SparseInternal.hpp:83-96)
    size_t total = (in.nnz + in.nr_cols) / 2;
    if (in.global_rows) {
        // a row slice: count what it holds on and below the diagonal
        total = 0;
        Triplet t;
        while (in.peek(t)) {
            if ((idx_t)(t.row + (idx_t) in.row_base) >= t.col) ++total;
            in.advance();
        }
        in.rewind();
    }
    size_t cnt = 0;
    idx_t row_start = (idx_t) in.row_base;
    for (size_t i = 0; i < nr; ++i) {
        size_t limit = (total - cnt) / (nr - i);
        PartitionSym *p = (i >= first && i < last) ? &parts[i - first] : nullptr;
        idx_t last_lower = 0;
        size_t diag = 0;
        size_t lower = take_partition_sym(in, row_start, limit, p, last_lower, diag);
        // The reference sizes the partition by its last row holding a lower
        // element and so loses trailing rows that only have a diagonal entry;
        // here every row that delivered its diagonal belongs to the partition.
        idx_t nrows = std::max<idx_t>(last_lower, (idx_t) diag);
        if (p) {
            Partition &lm = p->lower;
            lm.elems_size = lm.elems.size();
            lm.set_rowptr(lm.elems_size);
            lm.nnz = lower;
            lm.nr_rows = (size_t) nrows;
            lm.nr_cols = in.nr_cols;
            lm.row_start = row_start;
            lm.type = ENC_H;
        }
        PartBounds b;
        b.row_start = row_start;
        b.nr_rows = nrows;
        b.nnz = lower + diag;
        bounds.push_back(b);
        row_start += nrows;
        cnt += lower + diag;
    }
    if (cnt != total) {
        log_msg(LOG_ERR, "error in input matrix (matrix has less elements than "
                "claimed)\n");
        throw FatalError("element count mismatch");
    }
}

}  // namespace spx
