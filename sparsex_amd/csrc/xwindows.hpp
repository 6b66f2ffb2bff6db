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
//            LDS); entry 1 is spare.  Then XW_MAX windows: {first column, LDS offset | length << 16};
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
};

// `budget`: most doubles of unit windows per row-block (0: no windows at all -- the plan is then a
// plain copy); `gap`: intervals closer than this many doubles are merged.  General streams only
// (SPX_PASS_UNIT; symmetric read-once passes have their own LDS layout and are left alone).
void plan_unit_xwindows(const GpuStream &s, size_t ncols, uint32_t budget, uint32_t gap, XwPlan &plan,
                        unsigned nthreads);

}  // namespace spx
