// xwindows.hpp -- x windows for the unit passes of a row-block (device-side plan, derived from a
// finalized stream when it is uploaded; nothing of it is stored in the stream or in a saved file).
//
// The reference's unit bodies read x straight from memory, one element per nonzero
// (src/templates/horiz_tmpl.c:20-37, diag_tmpl.c:20-35, block_row_tmpl.c, ...: `x[x_indx + i * delta]`);
// its caches make the re-reads of neighbouring rows cheap.  On the GPU every lane of a unit pass
// gathers its W doubles of x through the vector L1 -- as many bytes as the values themselves,
// although a row-block only ever touches a few short stretches of x (a stencil row-block of R rows
// reads nine bands of R + 2 doubles).  Here those stretches become the row-block's *unit windows*:
// merged column intervals that the workgroup stages in LDS once, with coalesced 16-byte loads, and
// that every unit pass of the row-block then reads with ds_read -- no dependent global load, a
// quarter of the L1 requests.
//
// What the plan holds (all per device copy):
//   tab      XW_TAB entries of 8 bytes per row-block.  Entry 0: {lo | hi << 16, total}: the passes [lo, hi)
//            of the row-block are unit passes of width <= 4 with SPX_PASSF_XLDS (the longest run of
//            them) -- the kernel runs them as a software pipeline without looking at their headers first
//            -- and `total` is the length of the unit windows in doubles (the pass headers follow them in
//            LDS); entry 1: {first row, rows} of the row-block (every row-block, windows or not).  Then XW_MAX windows: {first column, LDS offset | length << 16};
//            length 0 ends the list.  Offsets and lengths in doubles, offsets even (16-byte LDS stores);
//            only the last window may have an odd length.
//   xdescs   a copy of GpuStream::descs in which `col0` of every unit of a row-block WITH windows
//            is the LDS offset (in doubles, from the start of the unit windows) of that column
//   passes   a copy of GpuStream::passes: unit passes of such row-blocks carry SPX_PASSF_XLDS, and
//            an inline descriptor (SPX_PASSF_INLINE) is translated like its entry in xdescs
// Row-blocks whose intervals do not fit (too many, or more doubles than the budget) keep absolute
// columns and gather through L2 as before: the kernel decides per pass.
#pragma once

#include "gpu_emit.hpp"

#include <cstdint>
#include <vector>

namespace spx {

constexpr uint32_t XW_TAB = 16;            // table entries per row-block (128 bytes: one load of a wavefront)
constexpr uint32_t XW_RANGES = 2;          // ... the first two hold the pass ranges of the pipeline
constexpr uint32_t XW_MAX = XW_TAB - XW_RANGES;   // unit windows per row-block
#define SPX_PASSF_XLDS 2u                  /* device-side flag of SpxPass::flags (never in a stream) */

struct XwEntry {
    uint32_t base;                         // first column
    uint32_t off_len;                      // LDS offset (bits 0-15) | length (bits 16-31), doubles
};

struct XwPlan {
    std::vector<XwEntry> tab;              // rbs.size() * XW_TAB
    std::vector<SpxUnitDesc> xdescs;       // descs.size()
    std::vector<SpxPass> passes;           // passes.size()
    uint32_t lds_doubles = 0;              // LDS of a launch: max over the row-blocks of y tile + leftover window + unit windows + pass headers
    size_t n_rb_windows = 0;               // row-blocks that got windows
    size_t n_rb_units = 0;                 // row-blocks that hold unit passes at all
    uint64_t staged_doubles = 0;           // doubles of x staged per product
    uint64_t unit_elems = 0, unit_elems_lds = 0;   // nonzeros in unit passes / of those, in passes that read LDS
    // leftover passes (SPX_PASS_GATHER) of row-blocks whose windows took their columns in as well: per pass
    // (index into `passes`) the first entry of `gdesc`, or UINT32_MAX; gdesc holds, per half of the pass' width
    // and per lane, two words: the LDS offsets of the lane's two columns (16 bits each) and its row | the
    // number of columns that are there << 16.  Used by the persistent kernel, which runs such a pass as
    // ceil(width / 2) passes of its pipeline.
    std::vector<uint32_t> gather_base;
    std::vector<uint32_t> gdesc;
};

// `budget`: most doubles of unit windows per row-block (0: no windows at all -- the plan is then a
// plain copy); `gap`: intervals closer than this many doubles are merged.  General streams only
// (SPX_PASS_UNIT; symmetric read-once passes have their own LDS layout and are left alone).
// `with_leftovers`: the columns of the row-block's leftover passes (SPX_PASS_GATHER) go into the windows as
// well where everything still fits (XwPlan::gather_base, gdesc: what the persistent kernel runs them from).
void plan_unit_xwindows(const GpuStream &s, size_t ncols, uint32_t budget, uint32_t gap, XwPlan &plan,
                        unsigned nthreads, bool with_leftovers = false);



// ---- persistent workgroups (csx_spmv_xwp_kernel, spmv_xwp_kernels.hip) ---------------------------------------
//
// The kernel above pays the fixed costs of a row-block -- two memory round trips before its first FMA, a
// barrier, the write-out -- once per workgroup, and a workgroup that waits adds nothing to the bytes in
// flight: on the bench matrix more than half of a workgroup's life is spent outside its pass loop.  The
// persistent form keeps a workgroup for the whole launch: it walks every G-th row-block of its XCD's
// part, holds two LDS regions (y tile + unit windows) and fills the one for the NEXT row-block while the
// passes of the current one run, and its wavefronts never drain their pipelines between row-blocks -- a
// wavefront executes ONE precompiled list of rounds (pairs of narrow unit passes, XwpRound) from its
// first row-block to its last.  The list is laid out here, per (workgroup, wavefront), from the same
// window plan; it depends on the launch geometry (workgroups per XCD, wavefronts per workgroup).
struct XwpRound {
    // per pass p = 0, 1: w[6p] | w[6p+1] << 32 = index of its first value in `values`; w[6p+2] = index of
    // lane 0's descriptor in the descriptor array; w[6p+3] | w[6p+4] << 32 = start mask (0: one
    // descriptor for all lanes); w[6p+5] = seg0 | nseg << 16 | width << 24.
    // (a half of a leftover pass: w[6p+2] = XwPlan::xdescs.size() + its first entry in XwPlan::gdesc -- on the
    // device the two arrays are one --, width = XWP_WIDTH_GATHER*.)
    // w[12], w[13]: SpxPass::elem0 of the two passes; w[14]: XWP_* flags; w[15]: row-block (index in rbs)
    uint32_t w[16];
};
#define XWP_WIDTH_GATHER2 5u  /* "width" of a pass of the list that is half a leftover pass: two nonzeros per lane  */
#define XWP_WIDTH_GATHER1 6u  /* ... or the odd last one of its width: one nonzero per lane                         */
#define XWP_LAST    1u      /* the wavefront's last round in this row-block: the row-block's end follows      */
#define XWP_GENERIC 2u      /* ... and the wavefront has passes outside the pipeline range to run there first */

struct XwpPlan {
    uint32_t waves = 4, wgs_per_xcd = 0, tail_rounds = 0;
    std::vector<XwpRound> rounds;          // the lists, one after the other (each ends with `tail_rounds` empty rounds)
    std::vector<uint64_t> stream_off;      // list of (workgroup b, wavefront w) starts at rounds[stream_off[b * waves + w]]
    std::vector<uint32_t> stream_len;      // ... and holds this many rounds (without the tail)
    uint32_t max_rows = 0, max_window = 0; // largest y tile / unit windows (doubles) of a row-block: the LDS regions
    uint64_t generic_passes = 0;           // passes that stay outside the pipeline
    bool usable = false;                   // false: a row-block needs what the persistent kernel does not do
                                           // (a leftover window in LDS, a row split over row-blocks), or more
                                           // than one pass in fifty stays outside the pipeline
};

// `first[9]`: the XCD parts of the launch (XcdSplit of spmv_device.hpp)
void plan_persistent_rounds(const GpuStream &s, const XwPlan &plan, const uint32_t first[9], uint32_t waves,
                            uint32_t wgs_per_xcd, uint32_t tail_rounds, XwpPlan &out, unsigned nthreads);
}  // namespace spx
