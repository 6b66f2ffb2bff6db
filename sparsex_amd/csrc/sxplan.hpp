// sxplan.hpp -- device-side pass headers for the pipelined read-once kernel (csx_spmv_sx_kernel,
// spmv_sx_kernels.hip); derived from a finalized stream when it is uploaded, nothing of it is stored in the
// stream or in a saved file.
//
// The reference's symmetric driver walks a unit with everything it needs at hand (ctl byte, x_indx, y_indx:
// src/templates/csx_sym_spmv_tmpl.c:60-106).  On the GPU a read-once pass (SPX_PASS_SYMSEG) was a chain of
// dependent memory round trips -- pass header, then the lanes' descriptors, then values and x -- with nothing
// in flight in between.  A pass whose lanes all belong to ONE unit (the emitter makes such passes for long
// runs: spx.gpu.sym_pure_passes) needs no descriptor load: its geometry fits the header.  Here every such
// pass of width <= 4 at the head of its row-block gets a device-side header that spells the geometry out
// for lane 0 -- lane l's segment is
//     row     = row_l0 + l * drow         (relative to the row-block)
//     columns = col_l0 + l * dcol ... + W - 1
//     slots   = slot_l0 + l * dcol ...    (or none)
// so that the kernel asks for a lane's values, x[row] and x[columns] in ONE round trip and runs the passes
// as a two-stage software pipeline.
//   words of an SX header (same 24 bytes as SpxPass; flag SPX_PASSF_SX in the flags byte):
//     w0  col_l0                     absolute column
//     w1  row_l0 [0,11) | drow [11,18) | dcol + 128 [18,26)
//     w2  val_off                    (unchanged)
//     w3  slot_l0, or SPX_NO_SLOT    (replaces rank0 | seg0 << 16)
//     w4  nseg | width << 8 | kind << 16 | flags << 24     (unchanged but for the flag)
//     w5  elem0                      (unchanged)
// All other passes keep their headers and run through the code of the plain kernels.
#pragma once

#include "gpu_emit.hpp"

#include <cstdint>
#include <vector>

namespace spx {

#define SPX_PASSF_SX 4u                    /* device-side flag of SpxPass::flags (never in a stream) */

struct SxPlan {
    std::vector<SpxPass> passes;           // passes.size(): the stream's headers, SX ones rewritten
    std::vector<uint32_t> n_sx;            // per row-block: its passes [0, n_sx) are SX passes
    size_t n_rb_sx = 0;                    // row-blocks with SX passes
    uint64_t sym_elems = 0, sx_elems = 0;  // nonzeros in read-once passes / of those, in SX passes
    uint64_t sx_passes = 0, sym_passes = 0;
};

// Symmetric streams with read-once segments only (others: an empty plan, n_sx all zero).
void plan_sym_pipeline(const GpuStream &s, SxPlan &plan, unsigned nthreads);

}  // namespace spx
