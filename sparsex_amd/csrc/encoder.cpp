// encoder.cpp -- see encoder.hpp.  Line references are to the reference's
// include/sparsex/internals/EncodingManager.hpp unless noted otherwise.
#include "encoder.hpp"

#include <algorithm>
#include <cassert>
#include <cmath>

namespace spx {

namespace {

// A row's free elements, given by their ascending positions in the current iteration order (1-based), read
// as maximal runs of equal steps: run = elements [first, first + count), each `step` behind its predecessor
// (the place in front of the row counts as position 0).  This is what the reference obtains by delta-
// encoding the positions in place and run-length-encoding the deltas (:457-466, :468-500); here the
// positions stay what they are and every consumer indexes them directly.
struct StepRun {
    size_t first, count;
    idx_t step;
};

bool next_step_run(const std::vector<idx_t> &pos, size_t at, StepRun &run)
{
    if (at >= pos.size()) return false;
    run.first = at;
    run.step = pos[at] - (at ? pos[at - 1] : 0);
    size_t end = at + 1;
    while (end < pos.size() && pos[end] - pos[end - 1] == run.step) ++end;
    run.count = end - at;
    return true;
}

// In a block order (blocks of `align` rows or columns, linearised one block column after the other) a run
// of step one is a band of consecutive linear positions; the element right in front of the run lies on
// the band too.  Cut at the block columns the band reads: `front` elements up to the first boundary,
// `whole` elements in whole block columns, `back` elements behind the last boundary (:1099-1123,
// :1207-1229, :1421-1440 compute the same three numbers from the run's start column).
struct Band {
    size_t first;                 // index of the band's first element (the run's, or the one in front of it)
    size_t front, whole, back;
};

Band align_band(const std::vector<idx_t> &pos, const StepRun &run, size_t align)
{
    Band b;
    b.first = run.first ? run.first - 1 : 0;
    size_t count = run.count + (run.first ? 1 : 0);
    const size_t into_column = (size_t) (pos[b.first] - 1) % align;
    b.front = into_column ? align - into_column : 0;
    count = count > b.front ? count - b.front : 0;
    b.back = count % align;
    b.whole = count - b.back;
    return b;
}

idx_t max_delta(const std::vector<idx_t> &xs)   // :408-424
{
    idx_t m = 0;
    for (size_t i = 1; i < xs.size(); ++i) m = std::max(m, xs[i] - xs[i - 1]);
    return m;
}

size_t delta_size(size_t v)   // Delta.hpp:35-48
{
    if (v <= 0xffu) return 1;
    if (v <= 0xffffu) return 2;
    if (v <= 0xffffffffu) return 4;
    return 8;
}

size_t iceil(size_t a, size_t b) { return a / b + (a % b != 0); }

}  // namespace

EncoderParams EncoderParams::from_config(const Config &cfg)
{
    EncoderParams p;
    p.min_limit = (size_t) cfg.get_long("spx.matrix.min_unit_size");
    p.max_limit = (size_t) cfg.get_long("spx.matrix.max_unit_size");
    p.min_coverage = cfg.get_double("spx.matrix.min_coverage");
    p.window_size = (size_t) cfg.get_long("spx.preproc.sampling.window_size");
    std::string m = cfg.get_str("spx.preproc.sampling");
    p.sampling = (m == "none") ? 0 : (m == "window") ? 1 : 2;
    p.min_cost = cfg.get_str("spx.preproc.heuristic") == "cost";
    p.portion = cfg.get_double("spx.preproc.sampling.portion");
    long ns = cfg.get_long("spx.preproc.sampling.nr_samples");
    p.split_blocks = cfg.get_bool("spx.matrix.split_blocks");
    p.onedim_blocks = cfg.get_bool("spx.matrix.onedim_blocks");
    p.nr_threads = cfg.nr_partitions();
    // parameter validation: CheckParams, :508-558
    if (p.sampling == 2) {
        if (ns <= 0) { log_msg(LOG_ERR, "invalid number of samples\n"); throw FatalError("samples"); }
        if (p.portion <= 0 || p.portion > 1) {
            log_msg(LOG_ERR, "invalid sampling portion\n");
            throw FatalError("portion");
        }
    } else if (p.sampling == 1) {
        if (ns <= 0) { log_msg(LOG_ERR, "invalid number of samples\n"); throw FatalError("samples"); }
        if (p.window_size <= 0) { log_msg(LOG_ERR, "invalid window size\n"); throw FatalError("window"); }
    }
    if (p.max_limit > (size_t) CTL_SIZE_MAX) p.max_limit = CTL_SIZE_MAX;
    if (p.max_limit < 1) p.max_limit = 1;
    p.samples_max = (size_t) std::max<long>(ns, 1);
    return p;
}

Encoder::Encoder(Partition *p, const EncoderParams &prm)
    : spm_(p), prm_(prm), sampling_enabled_(false),
      sort_window_size_(prm.window_size), samples_max_(prm.samples_max)
{
    ignore_all();
    if (prm_.sampling == 0) {
        sampling_enabled_ = false;
    } else {
        sampling_enabled_ = true;
        // the number of samples is per partition (:593-596)
        samples_max_ = (size_t) std::ceil((float) samples_max_ / (float) prm_.nr_threads);
        if (prm_.sampling == 2)
            sort_window_size_ =
                (size_t)(prm_.portion * (double) spm_->nnz / (double) samples_max_);
        compute_sort_splits();
        if (samples_max_ > sort_splits_.size()) samples_max_ = sort_splits_.size();
        select_splits();
    }
}

void Encoder::remove_ignore(int t)
{
    std::vector<int> types;
    enc_expand(t, types);
    for (int ty : types) {
        // one-dimensional blocks stay ignored unless explicitly enabled (:144-152)
        if (!prm_.onedim_blocks && (ty == ENC_BR1 || ty == ENC_BC1)) continue;
        ignore_.reset((size_t) ty);
    }
}

void Encoder::remove_ignore(const XformSeq &seq)
{
    for (const XformSpec &s : seq.seq) remove_ignore(s.type);
}

// ---- sampling windows (:1490-1599) ---------------------------------------------

void Encoder::compute_sort_splits()
{
    // One pass over the rows produces the sampling windows as (first row, end row, nonzeros): a
    // window ends with the row that brings it to sort_window_size_ nonzeros.  What is left at the
    // end joins the last window's count; it also extends that window to the last row unless it
    // holds more than half a window of nonzeros -- then it stays a row range of its own (with a
    // boundary, but without a count: the reference's vectors differ in length by two there).
    // A matrix smaller than one window is a single window (the reference reads back() of an
    // empty vector in that case).
    struct Window { size_t first, end, nnz; };
    std::vector<Window> win;
    const size_t rows = spm_->rowptr_size() - 1;
    Window cur{0, 0, 0};
    for (size_t r = 0; r < rows; ++r) {
        cur.nnz += (size_t)(spm_->rowptr[r + 1] - spm_->rowptr[r]);
        if (cur.nnz >= sort_window_size_) {
            cur.end = r + 1;
            win.push_back(cur);
            cur = Window{r + 1, 0, 0};
        }
    }
    bool tail_range = false;
    if (cur.nnz) {
        if (win.empty()) {
            win.push_back(Window{0, rows, cur.nnz});
        } else {
            win.back().nnz += cur.nnz;
            if (cur.nnz > sort_window_size_ / 2) tail_range = true;
            else win.back().end = rows;
        }
    }
    sort_splits_.push_back(0);
    for (const Window &w : win) {
        sort_splits_.push_back(w.end);
        sort_splits_nzeros_.push_back(w.nnz);
    }
    if (tail_range) sort_splits_.push_back(rows);
}

void Encoder::select_splits()
{
    // Which windows are sampled, as a formula of the slot number i.  With K boundaries and S
    // samples (S <= K): all of them when S == K; otherwise, when S exceeds K / 2, the first
    // `head` = K / 2 windows are taken as they are and only the remaining S - head samples are
    // spread, evenly (stride `skip`), over the remaining K - head -- written, as in the reference,
    // over the FIRST slots again, so a slot keeps its own number only behind the spread ones, and a
    // slot behind both is never assigned (zero here; uninitialised there).
    const size_t K = sort_splits_.size(), S = samples_max_;
    selected_splits_.assign(S, 0);
    const size_t head = (S != K && S > K / 2) ? K / 2 : 0;
    const size_t spread = S == K ? 0 : S - head;
    const size_t skip = S == K ? 0 : (K - head) / (spread + 1);
    for (size_t i = 0; i < S; ++i) {
        if (S == K) selected_splits_[i] = i;
        else if (i < spread) selected_splits_[i] = (i + 1) * skip;
        else if (i < head) selected_splits_[i] = i;
    }
}

// ---- statistics --------------------------------------------------------------------

void Encoder::generate_stats(Partition *sp, StatsCollection &stats)
{
    // Every live element of a row takes part, already encoded units included
    // (as a point at their anchor): the reference marks pattern members only
    // on temporary copies (:631-643, Element.hpp:138-151), so its InPattern
    // test never fires on the matrix itself.
    // (walked element by element: in column or diagonal order most "rows" of a sampling window are
    // empty, and an empty row has nothing to say)
    const size_t n = sp->elems_size;
    for (size_t j = 0; j < n;) {
        const idx_t row = sp->elems[j].row;
        for (; j < n && sp->elems[j].row == row; ++j) cols_buff_.push_back(sp->elems[j].col);
        update_stats(sp, cols_buff_, stats);
    }
}

void Encoder::generate_delta_stats(Partition *sp, StatsCollection &stats)
{
    // :647-705 -- with no element ever marked, each row is one candidate
    // delta unit; only its unit count is recorded (type is never ENC_NONE).
    std::vector<idx_t> xs;
    size_t nr = sp->rowptr_size() - 1;
    for (size_t i = 0; i < nr; ++i) {
        for (idx_t j = sp->rowptr[i]; j < sp->rowptr[i + 1]; ++j)
            xs.push_back(sp->elems[j].col);
        if (!xs.empty()) {
            size_t dsz = delta_size((size_t) max_delta(xs));
            size_t npatt = iceil(xs.size(), prm_.max_limit);
            size_t nnz = ((sizeof(idx_t) - dsz) * xs.size()) / sizeof(idx_t);
            if (sp->type == ENC_NONE)
                stats.append(Instantiation(sp->type, 0), StatsData(npatt, nnz, npatt));
            else
                stats.append(Instantiation(sp->type, 0), StatsData(0, 0, npatt));
            xs.clear();
        }
    }
}

void Encoder::update_stats(Partition *sp, std::vector<idx_t> &xs,
                           StatsCollection &stats)
{
    size_t align = (size_t) enc_block_align(sp->type);
    if (align) {
        update_stats_block(sp->type, xs, align, stats);
        return;
    }
    if (xs.empty()) return;

    // What a run of `len` equal steps is worth: units of at most max_limit elements; a last piece
    // shorter than min_limit is no unit and its elements do not count (:1376-1398)
    auto worth = [this](size_t len) {
        size_t units = len / prm_.max_limit, covered = len;
        const size_t last = len % prm_.max_limit;
        if (last && last >= prm_.min_limit) ++units;
        else covered -= last;
        return StatsData(covered, units);
    };
    // A run counts as `count` elements -- or count + 1 when the element in front of it is free to
    // join: there is one (this is not the row's first run) and the run before did not become a
    // unit itself (:1359-1367).  It must have more than one step and reach min_limit that way.
    bool at_row_start = true, prev_became_unit = false;
    StepRun run;
    for (size_t at = 0; next_step_run(xs, at, run); at = run.first + run.count) {
        const size_t len = run.count + ((!at_row_start && !prev_became_unit) ? 1 : 0);
        prev_became_unit = run.count > 1 && len >= prm_.min_limit;
        if (prev_became_unit) stats.append(Instantiation(sp->type, (size_t) run.step), worth(len));
        if (run.step != 0) at_row_start = false;
    }
    xs.clear();
}

void Encoder::update_stats_block(int type, std::vector<idx_t> &xs, size_t align,
                                 StatsCollection &stats)
{
    // every band of at least two whole block columns is a candidate block of that many columns (:1410-1487)
    StepRun run;
    for (size_t at = 0; next_step_run(xs, at, run); at = run.first + run.count) {
        if (run.step != 1) continue;
        const size_t columns = align_band(xs, run, align).whole / align;
        if (columns >= 2) stats.append(Instantiation(type, columns), StatsData(columns * align, 1));
    }
    xs.clear();
}

void Encoder::gen_all_stats(StatsCollection &stats)
{
    encoded_inst_.clear();
    if (sampling_enabled_ && spm_->rowptr_size() - 1 > samples_max_) {
        size_t samples_nnz = 0;
        spm_->transform(ENC_H);
        Partition window;
        for (size_t i = 0; i < samples_max_; ++i) {
            size_t sel = selected_splits_[i];
            // out of range in the reference (it reads past its vectors here)
            if (sel + 1 >= sort_splits_.size() || sel >= sort_splits_nzeros_.size()) break;
            size_t ws = sort_splits_[sel];
            size_t wsize = sort_splits_[sel + 1] - ws;
            // windows of at most one row end the sampling (:720-722)
            if (ws >= sort_splits_[sel + 1] - 1) break;
            spm_->get_window((idx_t) ws, (idx_t) wsize, window);
            if (window.nnz == 0) break;
            samples_nnz += sort_splits_nzeros_[sel];
            for (int t = ENC_H; t < ENC_MAX; ++t) {
                if (ignore_[(size_t) t]) continue;
                window.transform(t, false);
                generate_stats(&window, stats);
            }
            window.transform(ENC_H, false);
            spm_->put_window(window);
        }
        if (samples_nnz)
            stats.scale_all((double) spm_->nnz / (double) samples_nnz);
        if (prm_.split_blocks)
            stats.split_blocks(prm_.max_limit, spm_->nnz, prm_.min_coverage);
        stats.filter_coverage(spm_->nnz, prm_.min_coverage, encoded_inst_);
    } else {
        if (prm_.min_cost) generate_delta_stats(spm_, stats);
        for (int t = ENC_H; t < ENC_MAX; ++t) {
            if (ignore_[(size_t) t]) continue;
            spm_->transform(t);
            generate_stats(spm_, stats);
            if (enc_block_align(t) && prm_.split_blocks)
                stats.split_blocks(prm_.max_limit, spm_->nnz, prm_.min_coverage);
            stats.filter_coverage(spm_->nnz, prm_.min_coverage, encoded_inst_);
            if (prm_.min_cost) generate_delta_stats(spm_, stats);
        }
    }
}

unsigned long Encoder::type_score(int type, const StatsData &d) const
{
    // :836-861
    if (prm_.min_cost) {
        size_t nr_deltas = encoded_stats_.total.deltas + d.deltas;
        size_t nr_switches = (type == ENC_NONE) ? d.units : d.units + nr_deltas;
        if (d.nnz < d.units + nr_switches) return 0;
        return d.nnz - d.units - nr_switches;
    }
    return d.nnz - d.units;
}

int Encoder::choose_type(const StatsCollection &stats)
{
    int ret = ENC_NONE;
    unsigned long max_score = 0;
    for (auto &kv : stats.types) {
        unsigned long score = type_score(kv.first, kv.second.total);
        if (score == 0) {
            add_ignore(kv.first);
        } else if (score > max_score) {
            max_score = score;
            ret = kv.first;
        }
    }
    return ret;
}

// ---- encoding ---------------------------------------------------------------------------

Elem Encoder::make_unit(idx_t row, idx_t col, const val_t *vals, size_t size,
                        int type, size_t delta)
{
    if (size == 1) return make_single(row, col, vals[0]);   // Element.hpp:121-123
    Elem e;
    e.row = row; e.col = col; e.val = 0;
    e.voff = spm_->pool_alloc(vals, size);
    e.delta = (uint32_t) delta;
    e.size = (uint16_t) size;
    e.type = (uint8_t) type;
    e.pad_ = 0;
    e.pad2_ = 0;
    return e;
}

// Which blocks a band of `columns` whole block columns is cut into, as column counts in order (none: the band
// stays single elements).  Without spx.matrix.split_blocks: the band's own size must have been selected;
// it is cut into equal blocks no larger than max_limit elements, what does not fill one stays behind
// (:1125-1160).  With it: the selected sizes of this type, largest first, as often as each still fits
// (:1243-1262).
void Encoder::cut_band(int type, size_t columns, size_t align, std::vector<size_t> &blocks) const
{
    blocks.clear();
    if (columns < 2) return;
    if (!prm_.split_blocks) {
        if (!encoded_inst_.count(Instantiation(type, columns))) return;
        const size_t cap = prm_.max_limit / align, each = std::min(cap, columns);
        blocks.assign(std::max<size_t>(1, columns / cap), each);
        return;
    }
    for (auto it = encoded_inst_.rbegin(); it != encoded_inst_.rend(); ++it) {
        if (it->first != type) continue;
        for (; columns >= it->second; columns -= it->second) blocks.push_back(it->second);
    }
}

// Encodes one stretch of a row's free elements (positions ascending in the current order, with their
// values) into units of the selected instantiations and single elements, appended to `out`.
void Encoder::encode_stretch(idx_t row_no, std::vector<idx_t> &pos, std::vector<val_t> &vals, ElemSink &out)
{
    const int type = spm_->type;
    const size_t align = (size_t) enc_block_align(type);
    auto singles = [&](size_t from, size_t to) {
        for (size_t k = from; k < to; ++k) out.push_back(make_single(row_no, pos[k], vals[k]));
    };
    StepRun run;
    for (size_t at = 0; next_step_run(pos, at, run); at = run.first + run.count) {
        const size_t end = run.first + run.count;
        if (align) {
            // block orders: a band of step one, cut at its block columns, becomes blocks (:1085-1290)
            Band band = Band();
            if (run.step == 1) {
                band = align_band(pos, run, align);
                cut_band(type, band.whole / align, align, blocks_buff_);
            } else {
                blocks_buff_.clear();
            }
            if (blocks_buff_.empty()) {
                singles(run.first, end);
                continue;
            }
            // (the element in front of the run was written as a single element: the band takes it back)
            if (band.first != run.first) out.pop_back();
            size_t k = band.first + band.front;
            singles(band.first, k);
            for (const size_t columns : blocks_buff_) {
                out.push_back(make_unit(row_no, pos[k], &vals[k], columns * align, type, columns));
                k += columns * align;
            }
            singles(k, end);
            continue;
        }
        // linear orders: a run of a selected step becomes units of min_limit .. max_limit elements; a single
        // element in front of it -- not one that a unit of this stretch ends with -- joins the run (:1019-1082)
        size_t first = run.first, count = run.count;
        if (run.count != 1 && encoded_inst_.count(Instantiation(type, (size_t) run.step))) {
            if (first > 0 && !out.back().is_unit()) {
                out.pop_back();
                --first;
                ++count;
            }
            while (count >= prm_.min_limit) {
                const size_t take = std::min(prm_.max_limit, count);
                out.push_back(make_unit(row_no, pos[first], &vals[first], take, type, (size_t) run.step));
                first += take;
                count -= take;
            }
        }
        singles(first, first + count);
    }
    pos.clear();
    vals.clear();
}

void Encoder::encode_row(size_t row, ElemSink &newrow)
{
    // :1292-1319 -- stray elements between units are (re)encoded, units stay
    idx_t begin = spm_->rowptr[row], end = spm_->rowptr[row + 1];
    if (begin == end) return;
    idx_t row_no = spm_->elems[begin].row;
    for (idx_t j = begin; j < end; ++j) {
        const Elem e = spm_->elems[j];
        if (!e.is_unit()) {
            cols_buff_.push_back(e.col);
            vals_buff_.push_back(e.val);
            continue;
        }
        if (!cols_buff_.empty()) encode_stretch(row_no, cols_buff_, vals_buff_, newrow);
        newrow.push_back(e);
    }
    if (!cols_buff_.empty()) encode_stretch(row_no, cols_buff_, vals_buff_, newrow);
}

void Encoder::encode(int type)
{
    // :863-903
    if (type == ENC_NONE) return;
    spm_->transform(type);
    ElemSink out{spm_->elems.data(), 0};
    size_t nr = spm_->rowptr_size() - 1;
    for (size_t i = 0; i < nr; ++i) {
        encode_row(i, out);
        assert(out.size() <= (size_t) spm_->rowptr[i + 1]);
    }
    size_t n = out.size();
    spm_->elems_size = n;
    spm_->set_rowptr(n);
    add_ignore(type);
}

void Encoder::encode_all(std::ostream *log)
{
    if (!spm_->nnz) return;
    encoded_stats_.clear();
    enc_seq_.clear();
    // (the values of all units together cannot outnumber the nonzeros: address space now, pages as
    // they are written, no copy each time the pool doubles)
    if (spm_->nnz * sizeof(val_t) >= ((size_t) 32 << 20)) spm_->pool.reserve(spm_->pool.size() + spm_->nnz);
    for (;;) {
        StatsCollection type_stats;
        gen_all_stats(type_stats);
        if (log) *log << type_stats.to_string() << "\n";
        int type = choose_type(type_stats);
        if (type == ENC_NONE) break;
        if (log) *log << "Encode to " << enc_full_name(type) << "\n";
        encoded_stats_.append_type(type, type_stats);
        encode(type);
        enc_seq_.push_back(type);
    }
    spm_->transform(ENC_H);
    ElemVec().swap(spm_->scratch);
    if (log) {
        *log << "Encoding sequence: ";
        if (enc_seq_.empty()) *log << enc_full_name(ENC_NONE);
        for (size_t i = 0; i < enc_seq_.size(); ++i)
            *log << (i ? ", " : "") << enc_full_name(enc_seq_[i]);
        *log << "\n";
    }
}

void Encoder::encode_serial(const XformSeq &seq)
{
    if (!spm_->nnz) return;
    ignore_all();
    for (const XformSpec &s : seq.seq) {
        remove_ignore(s.type);
        for (size_t d : s.deltas) encoded_inst_.insert(Instantiation(s.type, d));
        encode(s.type);
        add_ignore(s.type);
        enc_seq_.push_back(s.type);
    }
    spm_->transform(ENC_H);
    ElemVec().swap(spm_->scratch);
}

}  // namespace spx
