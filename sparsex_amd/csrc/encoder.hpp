// encoder.hpp -- substructure mining, selection and encoding for one partition.
//
// Restates the algorithm of the reference's EncodingManager
// (include/sparsex/internals/EncodingManager.hpp): delta + run-length
// detection per iteration order, optional sampling windows, block splitting,
// coverage filter, score-based choice of the next type, encoding of the
// chosen type, repeated until no type scores.  Non-NUMA behaviour
// ("absorb the preceding stray element", EncodingManager.hpp:1036-1049,
// 1356-1367) is what is implemented, matching the reference's default build.
#pragma once

#include "config.hpp"
#include "partition.hpp"
#include "stats.hpp"

#include <bitset>
#include <set>
#include <sstream>

namespace spx {

struct EncoderParams {
    size_t min_limit = 4;        // spx.matrix.min_unit_size
    size_t max_limit = 255;      // spx.matrix.max_unit_size
    double min_coverage = 0.1;   // spx.matrix.min_coverage
    size_t window_size = 0;      // spx.preproc.sampling.window_size
    int sampling = 2;            // 0 none, 1 window, 2 portion
    bool min_cost = false;       // spx.preproc.heuristic == cost
    double portion = 0.01;       // spx.preproc.sampling.portion
    size_t samples_max = 48;     // spx.preproc.sampling.nr_samples
    bool split_blocks = true;    // spx.matrix.split_blocks
    bool onedim_blocks = false;
    size_t nr_threads = 1;       // spx.rt.nr_threads (samples are divided by it)

    static EncoderParams from_config(const Config &cfg);
};

class Encoder {
public:
    Encoder(Partition *p, const EncoderParams &prm);

    // ignore-set management (EncodingManager.hpp:123-179)
    void ignore_all() { ignore_.set(); }
    void add_ignore(int type) { ignore_.set((size_t) type); }
    void remove_ignore(int type_or_group);
    void remove_ignore(const XformSeq &seq);

    // automatic mode: mine, choose, encode until nothing scores (:906-960)
    void encode_all(std::ostream *log = nullptr);
    // explicit mode: encode exactly the requested (type, delta) pairs (:963-986)
    void encode_serial(const XformSeq &seq);

    // exposed for tests
    void gen_all_stats(StatsCollection &stats);
    int choose_type(const StatsCollection &stats);
    void encode(int type);
    const std::set<Instantiation> &encoded_inst() const { return encoded_inst_; }
    const std::vector<size_t> &sort_splits() const { return sort_splits_; }
    const std::vector<size_t> &selected_splits() const { return selected_splits_; }
    const std::vector<int> &encoding_sequence() const { return enc_seq_; }

private:
    void generate_stats(Partition *sp, StatsCollection &stats);
    void generate_delta_stats(Partition *sp, StatsCollection &stats);
    void update_stats(Partition *sp, std::vector<idx_t> &xs, StatsCollection &stats);
    void update_stats_block(int type, std::vector<idx_t> &xs, size_t align,
                            StatsCollection &stats);
    unsigned long type_score(int type, const StatsData &d) const;

    // Where an encoding round puts its elements: back into the partition's own array, behind
    // the position it reads at (a round never makes more elements than it has read).
    struct ElemSink {
        Elem *base;
        size_t n;
        void push_back(const Elem &e) { base[n++] = e; }
        void pop_back() { --n; }
        Elem &back() { return base[n - 1]; }
        size_t size() const { return n; }
    };
    void encode_row(size_t row, ElemSink &newrow);
    void encode_stretch(idx_t row_no, std::vector<idx_t> &pos, std::vector<val_t> &vals, ElemSink &out);
    void cut_band(int type, size_t columns, size_t align, std::vector<size_t> &blocks) const;
    Elem make_unit(idx_t row, idx_t col, const val_t *vals, size_t size,
                   int type, size_t delta);

    void compute_sort_splits();
    void select_splits();

    Partition *spm_;
    EncoderParams prm_;
    bool sampling_enabled_;
    size_t sort_window_size_;
    size_t samples_max_;
    std::vector<size_t> sort_splits_;
    std::vector<size_t> sort_splits_nzeros_;
    std::vector<size_t> selected_splits_;
    StatsCollection encoded_stats_;
    std::set<Instantiation> encoded_inst_;
    std::bitset<ENC_MAX> ignore_;
    std::vector<int> enc_seq_;
    std::vector<idx_t> cols_buff_;
    std::vector<val_t> vals_buff_;
    std::vector<size_t> blocks_buff_;     // block column counts of the band being cut (encode_stretch)
};

}  // namespace spx
