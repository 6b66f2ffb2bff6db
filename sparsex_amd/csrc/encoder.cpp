// encoder.cpp -- see encoder.hpp.  Line references are to the reference's
// include/sparsex/internals/EncodingManager.hpp unless noted otherwise.
#include "encoder.hpp"

#include <algorithm>
#include <cassert>
#include <cmath>

namespace spx {

namespace {

struct Rle { size_t freq; idx_t val; };

// in-place delta encoding; element 0 keeps its absolute value (:457-466)
void delta_encode(std::vector<idx_t> &xs)
{
    for (size_t i = xs.size() - 1; i > 0; --i) xs[i] -= xs[i - 1];
}

void rl_encode(const std::vector<idx_t> &in, std::vector<Rle> &out)
{
    out.clear();
    Rle r;
    r.freq = 1;
    r.val = in[0];
    for (size_t i = 1; i < in.size(); ++i) {
        if (in[i] != r.val) {
            out.push_back(r);
            r.freq = 1;
            r.val = in[i];
        } else {
            ++r.freq;
        }
    }
    out.push_back(r);
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

    std::vector<Rle> rles;
    delta_encode(xs);
    rl_encode(xs, rles);

    // What a run of `len` equal deltas is worth: units of at most max_limit elements; a last piece
    // shorter than min_limit is no unit and its elements do not count (:1376-1398)
    auto worth = [this](size_t len) {
        size_t units = len / prm_.max_limit, covered = len;
        const size_t last = len % prm_.max_limit;
        if (last && last >= prm_.min_limit) ++units;
        else covered -= last;
        return StatsData(covered, units);
    };
    // A run counts as `freq` elements -- or freq + 1 when the element in front of it is free to
    // join: there is one (this is not the row's first run) and the run before did not become a
    // unit itself (:1359-1367).  It must have more than one delta and reach min_limit that way.
    bool at_row_start = true, prev_became_unit = false;
    for (const Rle &run : rles) {
        const size_t len = run.freq + ((!at_row_start && !prev_became_unit) ? 1 : 0);
        prev_became_unit = run.freq > 1 && len >= prm_.min_limit;
        if (prev_became_unit) stats.append(Instantiation(sp->type, (size_t) run.val), worth(len));
        if (run.val != 0) at_row_start = false;
    }
    xs.clear();
}

void Encoder::update_stats_block(int type, std::vector<idx_t> &xs, size_t align,
                                 StatsCollection &stats)
{
    // :1410-1487 -- in a block iteration order a delta-1 run is a band of
    // full block columns once its start is aligned to the block boundary.
    if (xs.empty()) return;
    std::vector<Rle> rles;
    delta_encode(xs);
    rl_encode(xs, rles);

    idx_t unit_start = 0;
    for (const Rle &rle : rles) {
        unit_start += rle.val;
        if (rle.val == 1) {
            size_t nr_elem, skip_front;
            if (unit_start == 1) {
                skip_front = 0;
                nr_elem = rle.freq;
            } else {
                // the run really starts at the previous element
                skip_front = (size_t)(unit_start - 2) % align;
                if (skip_front != 0) skip_front = align - skip_front;
                nr_elem = rle.freq + 1;
            }
            if (nr_elem > skip_front) nr_elem -= skip_front;
            else nr_elem = 0;
            size_t other_dim = nr_elem / align;
            if (other_dim >= 2)
                stats.append(Instantiation(type, other_dim),
                             StatsData(other_dim * align, 1));
        }
        unit_start += rle.val * (idx_t)(rle.freq - 1);
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

void Encoder::do_encode(idx_t row_no, std::vector<idx_t> &xs,
                        std::vector<val_t> &vs, ElemSink &out)
{
    const int type = spm_->type;
    if (enc_is_block(type)) {
        if (!prm_.split_blocks) do_encode_block(row_no, xs, vs, out);
        else do_encode_block_alt(row_no, xs, vs, out);
        return;
    }

    // :1019-1082
    size_t vi = 0;
    std::vector<Rle> rles;
    delta_encode(xs);
    rl_encode(xs, rles);

    idx_t col = 0;
    for (const Rle &rle : rles) {
        size_t rle_freq = rle.freq;
        if (rle_freq != 1 &&
            encoded_inst_.count(Instantiation(type, (size_t) rle.val))) {
            size_t rle_start;
            col += rle.val;
            if (col != rle.val) {
                // not the first run of the row: take the stray element in
                // front of the run along, unless it belongs to a unit
                rle_start = (size_t) col;
                rle_freq = rle.freq;
                if (!out.back().is_unit()) {
                    rle_start -= (size_t) rle.val;
                    rle_freq++;
                    out.pop_back();
                    --vi;
                }
            } else {
                rle_start = (size_t) col;
                rle_freq = rle.freq;
            }
            while (rle_freq >= prm_.min_limit) {
                size_t curr = std::min(prm_.max_limit, rle_freq);
                out.push_back(make_unit(row_no, (idx_t) rle_start, &vs[vi], curr,
                                        type, (size_t) rle.val));
                vi += curr;
                rle_start += (size_t) rle.val * curr;
                rle_freq -= curr;
            }
            // leave col at the last element covered so far
            col = (idx_t) rle_start - rle.val;
        }
        for (size_t i = 0; i < rle_freq; ++i) {
            col += rle.val;
            out.push_back(make_single(row_no, col, vs[vi++]));
        }
    }
    assert(vi == vs.size());
    xs.clear();
    vs.clear();
}

void Encoder::do_encode_block(idx_t row_no, std::vector<idx_t> &xs,
                              std::vector<val_t> &vs, ElemSink &out)
{
    // :1085-1192 (split_blocks disabled)
    const int type = spm_->type;
    const size_t align = (size_t) enc_block_align(type);
    size_t vi = 0;
    std::vector<Rle> rles;
    delta_encode(xs);
    rl_encode(xs, rles);

    idx_t col = 0;
    for (const Rle &rle : rles) {
        size_t skip_front, skip_back, nr_elem;
        col += rle.val;
        if (col == 1) {
            skip_front = 0;
            nr_elem = rle.freq;
        } else {
            skip_front = (size_t)(col - 2) % align;
            if (skip_front != 0) skip_front = align - skip_front;
            nr_elem = rle.freq + 1;
        }
        if (nr_elem > skip_front) nr_elem -= skip_front;
        else nr_elem = 0;
        skip_back = nr_elem % align;
        if (nr_elem > skip_back) nr_elem -= skip_back;
        else nr_elem = 0;

        if (rle.val == 1 &&
            encoded_inst_.count(Instantiation(type, nr_elem / align)) &&
            nr_elem >= 2 * align) {
            size_t rle_start;
            if (col != 1) {
                rle_start = (size_t) col - 1;
                out.pop_back();
                --vi;
            } else {
                rle_start = (size_t) col;
            }
            for (size_t i = 0; i < skip_front; ++i)
                out.push_back(make_single(row_no, (idx_t)(rle_start + i), vs[vi++]));

            size_t max_limit = prm_.max_limit / align * align;
            size_t nr_blocks = nr_elem / max_limit;
            size_t nr_elem_block = std::min(max_limit, nr_elem);
            if (nr_blocks == 0) nr_blocks = 1;
            else skip_back += nr_elem - nr_elem_block * nr_blocks;

            for (size_t i = 0; i < nr_blocks; ++i) {
                out.push_back(make_unit(row_no,
                                        (idx_t)(rle_start + skip_front + i * nr_elem_block),
                                        &vs[vi], nr_elem_block, type,
                                        nr_elem_block / align));
                vi += nr_elem_block;
            }
            for (size_t i = 0; i < skip_back; ++i)
                out.push_back(make_single(
                    row_no,
                    (idx_t)(rle_start + skip_front + nr_elem_block * nr_blocks + i),
                    vs[vi++]));
        } else {
            for (size_t i = 0; i < rle.freq; ++i)
                out.push_back(make_single(row_no, col + (idx_t) i * rle.val, vs[vi++]));
        }
        col += rle.val * (idx_t)(rle.freq - 1);
    }
    assert(vi == vs.size());
    xs.clear();
    vs.clear();
}

void Encoder::do_encode_block_alt(idx_t row_no, std::vector<idx_t> &xs,
                                  std::vector<val_t> &vs, ElemSink &out)
{
    // :1194-1290 -- greedy cover of an aligned band with the accepted block
    // sizes, largest first
    const int type = spm_->type;
    const size_t align = (size_t) enc_block_align(type);
    size_t vi = 0;
    std::vector<Rle> rles;
    delta_encode(xs);
    rl_encode(xs, rles);

    idx_t col = 0;
    for (const Rle &rle : rles) {
        size_t skip_front, skip_back, nr_elem;
        col += rle.val;
        if (col == 1) {
            skip_front = 0;
            nr_elem = rle.freq;
        } else {
            skip_front = (size_t)(col - 2) % align;
            if (skip_front != 0) skip_front = align - skip_front;
            nr_elem = rle.freq + 1;
        }
        if (nr_elem > skip_front) nr_elem -= skip_front;
        else nr_elem = 0;
        skip_back = nr_elem % align;
        nr_elem -= skip_back;
        if (rle.val == 1 && nr_elem >= 2 * align) {
            size_t rle_start;
            if (col != 1) {
                rle_start = (size_t) col - 1;
                out.pop_back();
                --vi;
            } else {
                rle_start = (size_t) col;
            }
            for (size_t i = 0; i < skip_front; ++i)
                out.push_back(make_single(row_no, (idx_t)(rle_start++), vs[vi++]));

            size_t other_dim = nr_elem / align;
            for (auto it = encoded_inst_.rbegin(); it != encoded_inst_.rend(); ++it) {
                if (it->first != type) continue;
                while (other_dim >= it->second) {
                    size_t nr_elem_block = align * it->second;
                    out.push_back(make_unit(row_no, (idx_t) rle_start, &vs[vi],
                                            nr_elem_block, type, it->second));
                    rle_start += nr_elem_block;
                    vi += nr_elem_block;
                    nr_elem -= nr_elem_block;
                    other_dim -= it->second;
                }
            }
            skip_back += nr_elem;
            for (size_t i = 0; i < skip_back; ++i)
                out.push_back(make_single(row_no, (idx_t)(rle_start++), vs[vi++]));
        } else {
            for (size_t i = 0; i < rle.freq; ++i)
                out.push_back(make_single(row_no, col + (idx_t) i * rle.val, vs[vi++]));
        }
        col += rle.val * (idx_t)(rle.freq - 1);
    }
    assert(vi == vs.size());
    xs.clear();
    vs.clear();
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
        if (!cols_buff_.empty()) do_encode(row_no, cols_buff_, vals_buff_, newrow);
        newrow.push_back(e);
    }
    if (!cols_buff_.empty()) do_encode(row_no, cols_buff_, vals_buff_, newrow);
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
