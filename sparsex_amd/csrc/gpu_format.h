/*
 * gpu_format.h -- HBM layout of a tuned matrix: the row-block descriptor
 * stream walked by the HIP interpreter kernel (shared by host emitter and
 * device code; plain C structs, fixed widths).
 *
 * A *row-block* owns a contiguous range of rows and every nonzero that lands
 * in them.  CSX units produced by the preprocessor are re-tiled onto
 * row-blocks (multi-row units are cut at row-block borders), so a row-block
 * never writes y outside its rows.
 *
 * Inside a row-block every unit is expressed as a run of *row segments*: a
 * row segment is up to SPX_MAX_SEG_WIDTH consecutive columns of one row.
 *     dense r x c block        r segments of width c (wider blocks are cut
 *                              into column chunks); CSX block-row units,
 *                              column-major on the CPU, are transposed
 *     horizontal unit (d = 1)  chunks of SPX_HORIZ_CHUNK columns, same row
 *     vertical / diagonal /    one segment of width 1 per nonzero
 *     anti-diagonal / h (d>1)
 * Segment s of a unit sits at row0 + s*drow, columns col0 + s*dcol ... +W-1.
 * One 8-byte SpxUnitDesc describes the whole run (strides come from a 3-bit
 * kind and a 7-bit step; linear units with a step above 127 are rare and go
 * to the gather passes instead).
 *
 * Work is cut into *passes*: a pass is up to 64 row segments of the SAME
 * width W, one per lane of a wavefront.  Its values are stored interleaved
 * (pairs of columns, segment-minor) so that every load of a wavefront is one
 * contiguous block; a lane multiplies its W values with x[col..col+W-1] and
 * adds ONE partial sum to the row-block's y tile in LDS.  Segment-start bits
 * (one 64-bit mask per pass) tell a lane which descriptor it belongs to.
 *
 * The leftover nonzeros (CSX delta units) form *gather passes*: every row's
 * leftovers are cut into pieces of at most SPX_MAX_SEG_WIDTH nonzeros; a lane
 * owns one piece -- W values, W u16/u32 column offsets (relative to cbase,
 * element-major [W][nseg]), one u16 row -- and adds one partial sum, exactly
 * like a unit pass whose columns are not consecutive.  Where enough of a
 * row-block's leftovers have their columns close together, the workgroup stages
 * that window of x in LDS (coalesced loads) and those leftovers gather from LDS
 * (SPX_PASS_GATHER_LDS); far columns keep gathering through L2.
 *
 * The emitter (gpu_emit.cpp) shapes the segments for the lanes: nonzeros of
 * one-wide units that line up along their rows are re-cut into row segments,
 * and equal segments that follow a regular course (a diagonal, a stack, a
 * row) share one descriptor.  The pass headers of row-block i start at
 * passes[i * pass_stride] (fixed stride: a workgroup fetches its first headers
 * together with its row-block header).  Symmetric matrices: see
 * SPX_PASS_SYMTILE below.
 */
#ifndef SPX_GPU_FORMAT_H
#define SPX_GPU_FORMAT_H

#include <stdint.h>

#define SPX_MAX_TILE_SLOTS 3072   /* transposed-sum slots per row-block (LDS)          */
#define SPX_MAX_RB_ROWS    512    /* y tile of a row-block in LDS (doubles)      */
#define SPX_MAX_WIDE_ROWS  2048   /* ... of a wide row-block: up to four ordinary ones
                                     side by side that share their transposed-sum slots
                                     (streams with SPX_PASS_SYMSEG only).  A descriptor
                                     still holds a 9-bit row; its pass adds the first row
                                     of its part of the row-block (SpxPass::elem0)      */
#define SPX_MAX_WIDE_SLOTS 8192   /* transposed-sum slots of such a row-block (64 KB of LDS;
                                     with the y tile 80 KB: two workgroups per CU)        */
#define SPX_MAX_RB_ELEMS   8192   /* nonzeros per row-block (16-bit counters)    */
#define SPX_MAX_SEG_WIDTH  8      /* columns per row segment                      */
#define SPX_HORIZ_CHUNK    8      /* horizontal units are cut into such chunks   */
#define SPX_PASS_SEGS      64     /* row segments (lanes) per unit pass           */

#define SPX_PASS_UNIT   0
#define SPX_PASS_SYMTILE 3   /* symmetric path, whole matrix in one process: up to 8
                                dense 8x8 tiles of the stored lower triangle, lanes
                                8t..8t+7 = the rows of tile t, W = 8.  Each value is
                                read ONCE and used twice: the lane's row sum goes to
                                the y tile as usual, the tile's column sums (reduced
                                over its 8 lanes in registers) go to the row-block's
                                transposed-sum slots.  One descriptor per tile:
                                col0, bits = row0 (9 bits) | first slot << 9         */

#define SPX_PASS_GATHER 2    /* leftover nonzeros as row pieces: lane l owns up to
                                SPX_MAX_SEG_WIDTH nonzeros of ONE row with explicit
                                column offsets; values interleaved like a unit
                                pass, offsets element-major [W][nseg].  The row-
                                block's u16 `segrows` entry of a piece holds its row
                                (bits 0-10) and its length - 1 (bits 11-13): a pass is
                                as wide as its longest piece, shorter ones are padded */
#define SPX_SEGROW(row, len) ((uint16_t)((row) | (((len) - 1u) << 11)))
#define SPX_SEGROW_ROW(sr)   ((uint32_t)(sr) & 2047u)
#define SPX_SEGROW_LEN(sr)   ((((uint32_t)(sr) >> 11) & 7u) + 1u)

#define SPX_PASS_GATHER_LDS 4 /* the same, for leftovers whose columns fall into the
                                row-block's x window: the workgroup stages
                                x[xwin_base, xwin_base + xwin_len) in LDS once,
                                coalesced, and the lanes gather from there; u16
                                offsets relative to xwin_base                      */
#define SPX_MAX_XWIN    4096  /* doubles of x a row-block may stage in LDS (32 KB)  */

#define SPX_PASS_SYMSEG 5     /* symmetric path, values read ONCE: row segments of the
                                stored lower triangle, one per lane as in a unit pass;
                                besides its row sum a(r,c..c+W-1) . x[c..] the lane adds
                                the W transposed products a(r,c+w) * x[r] to the
                                row-block's transposed-sum slots (the same slots the
                                tiles use).  Two SpxUnitDesc entries per unit: the unit
                                descriptor, then {slot of segment 0's first column, 0};
                                segment s uses slot + s * dcol.  Slot 0xFFFFFFFF: no slot
                                (window full, or the segment straddles the row-block's
                                first row): the lane adds straight to y with global
                                atomics.  Only in streams whose sums are handed over
                                atomically (sym_atomic)                              */
#define SPX_NO_SLOT 0xFFFFFFFFu

#define SPX_KIND_BLOCK  0u   /* rows of a dense block: drow 1, dcol 0             */
#define SPX_KIND_HORIZ  1u   /* same row, column step `step`                      */
#define SPX_KIND_VERT   2u   /* same column, row step `step`                      */
#define SPX_KIND_DIAG   3u   /* row and column step `step`                        */
#define SPX_KIND_ADIAG  4u   /* row step `step`, column step -`step`              */
#define SPX_MAX_STEP    127u

typedef struct {
    uint32_t col0;       /* first column of segment 0 (0-based, absolute)       */
    uint32_t bits;       /* [0,9) row of segment 0 relative to the row-block
                            [9,22) segments of this row-block in front of it
                            [22,25) SPX_KIND_*      [25,32) step                 */
} SpxUnitDesc;           /* 8 bytes */

static inline uint32_t spx_desc_bits(uint32_t row0, uint32_t sstart, uint32_t kind,
                                     uint32_t step)
{
    return (row0 & 511u) | ((sstart & 8191u) << 9) | ((kind & 7u) << 22) | ((step & 127u) << 25);
}

#define SPX_PASSF_INLINE 1u  /* unit pass whose lanes all belong to ONE descriptor (its start mask would be
                                zero): `mask` holds that descriptor instead, {col0, bits}.  The wavefront has
                                it with the pass header -- which is fetched a round ahead -- and computes rows
                                and x addresses without waiting for a descriptor load: one dependent round
                                trip per pass instead of two (the descriptor is in `descs` as well, where the
                                host-side decoders keep reading it)                                      */
typedef struct {
    uint64_t mask;       /* unit pass, bit l: lane l's segment starts a new unit;
                            bit 0 is never set.  SPX_PASSF_INLINE: the pass' only
                            descriptor, col0 | (uint64_t) bits << 32               */
    uint32_t val_off;    /* first value of the pass, relative to the row-block  */
    uint16_t rank0;      /* unit pass: descriptor of lane 0's segment           */
    uint16_t seg0;       /* unit pass: segments in front of lane 0
                            gather pass: row pieces in front (index into the
                            row-block's u16 rows at seg_off)                     */
    uint8_t  nseg;       /* active lanes, 1..64                                  */
    uint8_t  width;      /* W: nonzeros per lane                                 */
    uint8_t  kind;       /* SPX_PASS_UNIT / _GATHER / _GATHER_LDS / _SYMTILE      */
    uint8_t  flags;      /* SPX_PASSF_*                                          */
    uint32_t elem0;      /* gather pass: leftover nonzeros of the row-block in
                            front of this pass (index of its first column offset)
                            other passes: added to the rows of their descriptors
                            (0 but in wide row-blocks, SPX_MAX_WIDE_ROWS)           */
} SpxPass;               /* 24 bytes */

typedef struct {
    uint64_t val_off;     /* first value of the row-block in values[]          */
    uint32_t pass_off;    /* first SpxPass                                      */
    uint32_t desc_off;    /* first SpxUnitDesc                                  */
    uint32_t cidx_off;    /* offset of the leftovers' column offsets, in units of
                             16 bytes (up to 64 GB of them)                      */
    uint32_t seg_off;     /* first u16 row of the leftover row pieces           */
    uint32_t cbase;       /* column base of the leftovers                       */
    uint32_t row0;        /* first row owned (global)                           */
    uint16_t n_rows;      /* rows owned                                         */
    uint16_t n_pass;
    uint8_t  cidx_width;  /* bytes per column offset: 2, 4, or 3 = a u16 array (low
                             halves) followed, at hi_off, by a u8 array (bits 16-23) */
    uint8_t  flags;       /* SPX_RB_* */
    uint16_t n_slots;     /* transposed-sum slots (SPX_PASS_SYMTILE): one per
                             distinct column in front of row0 that a tile of this
                             row-block touches; they sit in LDS in front of the y
                             tile, so that slot n_slots + i is owned row i        */
    uint32_t carry_slot;  /* SPX_RB_SHARED: slot of the partial sum             */
    uint32_t spill_off;   /* the n_slots sums go to spill[spill_off ...]; a second
                             kernel adds them to the rows they belong to          */
    uint32_t xwin_base;   /* first column of the x window staged in LDS          */
    uint16_t xwin_len;    /* its length in doubles (0: none), <= SPX_MAX_XWIN    */
    uint16_t near_off;    /* u16 offsets of the SPX_PASS_GATHER_LDS passes start
                             at cidx_off * 16 + near_off * 16 bytes               */
    uint32_t hi_off;      /* cidx_width 3: the u8 array starts at
                             cidx_off * 16 + hi_off * 16 bytes                    */
    uint32_t pad2_;
} SpxRowBlock;            /* 64 bytes */

#define SPX_RB_SHARED 1u  /* owns one chunk of an over-long row; the partial
                             goes to carry[carry_slot] and a fix-up kernel sums */
#define SPX_RB_PHASE_START 4u /* general path, column phases (spx.gpu.col_phases): the stream holds the
                             matrix as a sum of column slices, A = A_0 + A_1 + ..., each slice a run of
                             row-blocks of its own, launched one after the other: slice k > 0 adds to
                             what the slices in front of it stored (beta = 1), and a workgroup only
                             ever gathers x from one slice of the columns -- a slice that fits the 4 MB
                             of L2 of an XCD.  This flag marks the first row-block of a slice k > 0  */
#define SPX_RB_ACCUM 8u   /* ... or (spx.gpu.col_phases = c2 | c4 | c8) all slices run in ONE launch, slice k on
                             its own group of 8 / K XCDs, so that an XCD's L2 only ever sees 1 / K of x: every
                             row-block then ADDS its y tile to y (global atomics, coalesced by rows) on top of
                             a pass that put beta * y there.  Set on every row-block of such a stream        */
#define SPX_RB_PRIVATE 2u /* symmetric path, atomic hand-over: nobody else adds to the rows of
                             this row-block (no slot group of any row-block, no slot-less
                             read-once segment, no mirror list reaches them), so it STORES them,
                             y = alpha * (sums + d * x) + beta * y, and csx_sym_init_kernel leaves
                             them out: no init pass and no read-modify-write for these rows  */

/* rows split over several row-blocks: y[row] = sum of carry[first..first+n) */
typedef struct {
    uint32_t row;
    uint32_t first_slot;
    uint32_t n_slots;
} SpxSharedRow;

/* Position of value (segment lane, column w) inside a unit pass of nseg
 * segments of width W: columns are stored in pairs so that a lane reads 16
 * bytes at a time; an odd last column is stored alone. */
static inline uint32_t spx_pass_value_index(uint32_t lane, uint32_t w, uint32_t nseg,
                                            uint32_t width)
{
    uint32_t pair = w >> 1;
    if ((width & 1u) && w == width - 1u) return pair * 2u * nseg + lane;
    return pair * 2u * nseg + lane * 2u + (w & 1u);
}

/* the start mask of a unit pass (an inline descriptor stands for "no starts") */
static inline uint64_t spx_pass_mask(const SpxPass *ps)
{
    return (ps->flags & SPX_PASSF_INLINE) ? 0ull : ps->mask;
}

#endif /* SPX_GPU_FORMAT_H */
