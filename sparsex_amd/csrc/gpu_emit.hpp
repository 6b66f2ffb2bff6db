// gpu_emit.hpp -- re-tiles encoded CSX partitions into the row-block
// descriptor stream of gpu_format.h (host side; the arrays are uploaded to
// HBM as they are).
//
// Plays the role CsxManager::MakeCsx plays for the CPU format in the
// reference (include/sparsex/internals/CsxManager.hpp:301-437): it walks the
// encoded partition and lays units and values out for the executor.
#pragma once

#include "gpu_format.h"
#include "partition.hpp"

#include <utility>
#include <vector>

namespace spx {

// A dense 8x8 tile of the strictly lower triangle (symmetric path).
struct SymTile {
    idx_t row0, col0;      // 0-based, global
    val_t v[64];           // row-major
};

// A row segment of the strictly lower triangle that is read once (SPX_PASS_SYMSEG).
struct SymSeg {
    idx_t row, col;        // 0-based, global
    uint8_t width;         // 3..8
    val_t v[8];
};
typedef std::vector<SymSeg, BigAlloc<SymSeg>> SymSegVec;   // (ten gigabytes on the contract matrix)

struct GpuStream {
    ValVec values;                // (6 GB on the contract matrix: big_alloc.hpp)
    std::vector<SpxUnitDesc> descs;
    std::vector<SpxPass> passes;
    std::vector<uint8_t> cidx;
    std::vector<uint16_t> segrows;
    std::vector<SpxRowBlock> rbs;
    std::vector<SpxSharedRow> shared;
    uint32_t n_carry = 0;
    // symmetric path: diagonal of the rows covered by the emitted partitions,
    // indexed by global row (zero elsewhere)
    std::vector<val_t> dvalues;
    // symmetric path, whole matrix in this process: every row has a row-block
    // and the diagonal term is added in the kernel's write-out (no init pass)
    bool sym_fused = false;
    // symmetric path with tiles (SPX_PASS_SYMTILE): column of every spill slot
    // (host side only) and, per row, the slots whose sums belong to it
    std::vector<uint32_t> spill_col;
    std::vector<uint32_t> fix_ptr, fix_idx;
    // first column of every group of eight slots (tiles start on columns that are
    // multiples of eight); slot i of the stream belongs to column
    // slot_group_col[i / 8] + i % 8
    std::vector<uint32_t> slot_group_col;
    // host side, emission only: {first column, width} of read-once segments that found no slot
    // and add straight to y (what mark_private_rowblocks needs to know besides the slots)
    std::vector<std::pair<uint32_t, uint32_t>> direct_cols;
    // symmetric slice: mirror-image nonzeros that land, thinly spread, on rows in front of
    // the own rows (a stencil matrix's constraint couplings seen from the last process) are
    // not worth row-blocks of their own (a workgroup per handful of nonzeros): they are kept
    // as a small CSR over those rows, one thread per row adds them (csx_sym_mirror_rows_kernel)
    std::vector<uint32_t> mirror_rows;     // global row of every such row, ascending
    std::vector<uint32_t> mirror_ptr;      // mirror_rows.size() + 1
    std::vector<uint32_t> mirror_col;      // column = own row the value multiplies x of
    std::vector<val_t> mirror_val;
    // symmetric tiles: how the transposed sums reach their rows -- false: spilled
    // and collected per row by a second kernel in a fixed order; true: added
    // straight into y with 64-byte groups of global_atomic_add_f64
    bool sym_atomic = false;
    bool deterministic = false;   // spx.gpu.deterministic: per-wavefront y tiles, summed in order
    bool wave_tiles = false;      // per-wavefront y tiles (chosen by the launch autotuner or forced)
    uint32_t n_spill = 0;                      // spill slots (= spill_col.size() after emission)
    uint32_t lds_doubles = SPX_MAX_RB_ROWS;    // largest n_slots + n_rows
    // pass headers of row-block i start at passes[i * pass_stride] once
    // finalize_stream() ran (0: packed, as the emitter appends them)
    uint32_t pass_stride = 0;
    uint32_t waves = 4;           // wavefronts per workgroup the kernel is launched with
    bool band_order = false;      // spx.gpu.band_order: launch order by strips across recurring bands of x (device side only)
    bool arena = false;           // spx.gpu.arena: all arrays of the stream in one HBM allocation (device side only)
    // spx.gpu.unit_windows (device side only, xwindows.hpp): most doubles of x a row-block may stage in
    // LDS for its unit passes (0: none planned), intervals closer than `xw_gap` doubles are merged, and
    // whether the product starts out using them (the launch tuner measures both)
    uint32_t xw_budget = 0, xw_gap = 16;
    bool xw_on = false;
    // spx.gpu.sym_pipeline (device side only, sxplan.hpp): plan the read-once pipeline at upload, and whether
    // the product starts out using it (the launch tuner measures both)
    bool sx_plan = false, sx_on = false;
    // accounting
    size_t nnz_stored = 0;        // nonzeros held in `values` (without padding)
    size_t n_unit_elems = 0;
    size_t n_delta_elems = 0;
    size_t n_units = 0;

    size_t n_pass_used() const
    {
        size_t n = 0;
        for (const SpxRowBlock &rb : rbs) n += rb.n_pass;
        return n;
    }
    size_t index_bytes() const
    {
        return descs.size() * sizeof(SpxUnitDesc) + n_pass_used() * sizeof(SpxPass) + cidx.size() +
               (sym_atomic ? slot_group_col.size() : fix_ptr.size() + fix_idx.size()) * 4 +
               (mirror_rows.size() + mirror_ptr.size() + mirror_col.size()) * 4 +
               segrows.size() * 2 + rbs.size() * sizeof(SpxRowBlock);
    }
};

struct GpuEmitParams {
    size_t target_elems = 2048;   // spx.gpu.rowblock_elems
    size_t max_rows = SPX_MAX_RB_ROWS;   // spx.gpu.rowblock_rows (<= SPX_MAX_RB_ROWS)
    size_t sym_min_run = 2;              // spx.gpu.sym_segment_min: shortest run of columns read once
    size_t sym_max_run = SPX_MAX_SEG_WIDTH;   // spx.gpu.sym_segment_max: widest read-once segment a run is cut into
    size_t wide_rows = SPX_MAX_RB_ROWS;  // spx.gpu.sym_wide_rows: rows of a row-block made of several
                                         // planned ones (read-once segments only; <= SPX_MAX_WIDE_ROWS)
    bool skip_empty = false;      // accumulate mode: rows without nonzeros need no write
    bool stack_segments = true;   // spx.gpu.stack_segments: equal row segments of consecutive
                                  // rows share one descriptor as a dense block
    bool recut_linear = true;     // spx.gpu.recut_linear: nonzeros of vertical / diagonal /
                                  // strided units that line up along their rows run as row segments
    bool inline_desc = true;      // spx.gpu.inline_desc: SPX_PASSF_INLINE
    bool keep_units = true;       // spx.gpu.keep_units: ... but a mined unit none of whose nonzeros has a
                                  // neighbour along its row stays the unit it is (one descriptor)
    bool sym_remine = true;       // spx.gpu.sym_remine (see append_sym_expanded)
    bool x_window = true;         // spx.gpu.x_window: stage a window of x in LDS for leftovers
                                  // whose columns lie close together
    int sym_segments = -1;        // spx.gpu.sym_segments: 1 / 0, -1 = where most of the lower triangle
                                  // lies in runs of three and more columns
    bool sym_pure_passes = true;  // spx.gpu.sym_pure_passes: long runs of read-once segments fill passes of
                                  // their own (one descriptor, in the header: csx_spmv_sx_kernel pipelines them)
    bool sym_once = true;         // spx.gpu.sym_once: dense 8x8 tiles of a symmetric matrix are
                                  // read once (one process holding the whole matrix only)
    const std::vector<SymTile> *tiles = nullptr;   // symmetric, fused: tiles read once (sorted by row0)
    const SymSegVec *symsegs = nullptr;  // symmetric: lower row segments read once (sorted by row)
};

// Appends the row-blocks of partition `p` (horizontal order) to `out`.
// Rows are numbered globally (p.row_start + local row).  With nthreads > 1
// runs of row-blocks are built concurrently and joined in order (same stream).
void emit_gpu(const Partition &p, const GpuEmitParams &prm, GpuStream &out, unsigned nthreads = 1);

// Joins `src` to the end of `dst` (neither finalized yet): what per-partition /
// per-range emitter threads produced becomes one stream.
void append_stream(GpuStream &dst, GpuStream &&src, uint64_t values_placed_at = UINT64_MAX);

// Lays the pass headers out at a fixed stride per row-block (the largest pass
// count), so that a workgroup can fetch its first headers without waiting for
// its row-block header: one dependent memory round trip less per workgroup.
// Only the used entries are ever read.  Call once, after the last emit_gpu().
void finalize_stream(GpuStream &s, size_t nrows);

// Symmetric path, after finalize_stream: flags the row-blocks whose rows receive nothing from
// outside (SPX_RB_PRIVATE, see gpu_format.h).  Only row-blocks inside [own_lo, own_hi) qualify.
void mark_private_rowblocks(GpuStream &s, size_t nrows, idx_t own_lo, idx_t own_hi);

// Symmetric path: the strictly lower triangle held by `lower` (rows local to
// the partition) plus its mirror image, as one general partition in global
// row numbering (row_start 0).  Appends to `out`.  The mirrored nonzeros are
// either re-cut into row segments and dense blocks (`remine_upper`, the
// default: what the kernel's lanes want) or mirrored unit by unit
// (horizontal <-> vertical, diagonal and anti-diagonal stay, block-row
// R x c <-> block-col c x R).
void append_sym_expanded(const Partition &lower, Partition &out, bool remine_upper = true);

// Symmetric path, values read once where it pays.  The strictly lower triangle
// held by `lowers` (rows local to each partition) is turned into one general
// partition per row range of `ranges` (ascending, contiguous, 0-based global
// [lo, hi); the rows of every partition of `lowers` must be one of the ranges,
// further ranges in front take the mirror image that lands on rows of other
// processes; where that image is thin -- fewer than 128 nonzeros in a stretch of 512 rows --
// it goes to `sparse_mirror` instead, if given): dense 8x8 tiles on rows that are multiples of eight go to
// `tiles[range]` (sorted by row), everything else goes to `outs[range]` together
// with the mirror image that falls into the range, both re-cut into row
// segments and blocks; rows of outs[j] are relative to ranges[j].lo.  With `symsegs`, runs of three
// and more consecutive columns of a row of what is not in a tile are not mirrored either: they go to
// symsegs[range] (sorted by row) and are read once as well (SPX_PASS_SYMSEG).  Ranges
// are independent of each other, so they are built -- and can then be emitted
// -- concurrently.
struct SymRange { idx_t lo, hi; };
struct MirrorPoint { idx_t row, col; val_t val; };      // 0-based global; row < every own row
void build_sym_ranges(const std::vector<Partition> &lowers, const std::vector<SymRange> &ranges,
                      bool want_tiles, std::vector<Partition> &outs,
                      std::vector<std::vector<SymTile>> &tiles, unsigned nthreads,
                      std::vector<MirrorPoint> *sparse_mirror = nullptr,
                      std::vector<SymSegVec> *symsegs = nullptr, size_t min_run = 2, size_t max_run = SPX_MAX_SEG_WIDTH);

// coordinates (1-based, horizontal order) of element k of a unit
inline void unit_elem_coords(const Elem &u, size_t k, idx_t &r, idx_t &c)
{
    const idx_t d = (idx_t) u.delta;
    switch (u.type) {
    case ENC_H:  r = u.row;               c = u.col + (idx_t) k * d; return;
    case ENC_V:  r = u.row + (idx_t) k * d; c = u.col;               return;
    case ENC_D:  r = u.row + (idx_t) k * d; c = u.col + (idx_t) k * d; return;
    case ENC_AD: r = u.row + (idx_t) k * d; c = u.col - (idx_t) k * d; return;
    default: break;
    }
    const idx_t a = (idx_t) enc_block_align(u.type);
    if (enc_is_block_row(u.type)) {        // a rows x delta cols, column-major
        r = u.row + (idx_t) k % a;
        c = u.col + (idx_t) k / a;
    } else {                               // delta rows x a cols, row-major
        r = u.row + (idx_t) k / a;
        c = u.col + (idx_t) k % a;
    }
}

}  // namespace spx
