/*
 * gpu_format.h -- HBM layout of a tuned matrix: the row-block descriptor
 * stream walked by the HIP interpreter kernel (shared by host emitter and
 * device code; plain C structs, fixed widths).
 *
 * A *row-block* owns a contiguous range of rows and every nonzero that lands
 * in them.  CSX units produced by the preprocessor are re-tiled onto
 * row-blocks (multi-row units are cut at row-block borders), so a row-block
 * never writes y outside its rows.  Inside a row-block the nonzeros are
 * stored in two regions:
 *
 *   unit region   values of the substructure units, in descriptor order; one
 *                 16-byte SpxUnitDesc per unit gives anchor and strides, and
 *                 element k of the unit sits at
 *                    linear unit (mod == 0): row0 + k*drow,  col0 + k*dcol
 *                       (horizontal: drow 0; vertical: dcol 0; diagonal:
 *                        drow = dcol; anti-diagonal: dcol = -drow)
 *                    dense block (mod == c): row0 + k / c,   col0 + k % c
 *                       (row-major r x c; the CPU format's column-major
 *                        block-row units are transposed when emitted)
 *   delta region  the leftover nonzeros, row-major; one u16 row per row
 *                 segment and one column offset (u16 or u32, relative to
 *                 cbase) per nonzero -- the GPU form of CSX delta units
 *
 * Both regions carry one "segment start" bit per nonzero (a new unit / a new
 * row segment begins here); lanes rank those bits to find their descriptor.
 * A region is cut into passes of SPX_PASS_ELEMS nonzeros (64 lanes x
 * SPX_LANE_ELEMS consecutive nonzeros); pass_rank[] holds the number of
 * segment starts in front of each pass, which makes passes independent: one
 * workgroup owns a row-block and its wavefronts take the passes in turn.
 */
#ifndef SPX_GPU_FORMAT_H
#define SPX_GPU_FORMAT_H

#include <stdint.h>

#define SPX_LANE_ELEMS   4      /* consecutive nonzeros per lane and pass      */
#define SPX_PASS_ELEMS   256    /* 64 * SPX_LANE_ELEMS                         */
#define SPX_PASS_WORDS   8      /* u32 words of start bits per pass            */
#define SPX_MAX_RB_ROWS  512    /* y tile per wavefront in LDS (doubles)       */
#define SPX_MAX_RB_ELEMS 8192   /* nonzeros per region (estart is 16 bit)      */

typedef struct {
    uint32_t col0;       /* anchor column (0-based, absolute)                  */
    int32_t  dcol;       /* linear: column stride per element                  */
    uint16_t estart;     /* first nonzero of the unit inside the unit region   */
    uint16_t row0;       /* anchor row relative to the row-block               */
    int16_t  drow;       /* linear: row stride per element                     */
    uint8_t  mod;        /* dense block: row length c; 0 = linear unit         */
    uint8_t  pad_;
} SpxUnitDesc;           /* 16 bytes */

typedef struct {
    uint64_t val_off;     /* first value of the row-block in values[]          */
    uint32_t desc_off;    /* first SpxUnitDesc                                  */
    uint32_t bits_off;    /* first u32 word of start bits (unit passes first);
                             bits_off / SPX_PASS_WORDS indexes pass_rank[]      */
    uint32_t cidx_off;    /* byte offset of the delta region's column offsets  */
    uint32_t seg_off;     /* first u16 row of the delta region's row segments  */
    uint32_t cbase;       /* column base of the delta region                    */
    uint32_t row0;        /* first row owned (relative to the partition slice) */
    uint16_t n_rows;      /* rows owned                                         */
    uint16_t n_unit_elems;
    uint16_t n_delta_elems;
    uint8_t  cidx_width;  /* 2 or 4 bytes per column offset                     */
    uint8_t  flags;       /* SPX_RB_* */
    uint32_t carry_slot;  /* SPX_RB_SHARED: slot of the partial sum             */
    uint32_t pad_;
} SpxRowBlock;            /* 48 bytes */

#define SPX_RB_SHARED 1u  /* owns one chunk of an over-long row; the partial
                             goes to carry[carry_slot] and a fix-up kernel sums */

/* rows split over several row-blocks: y[row] = sum of carry[first..first+n) */
typedef struct {
    uint32_t row;
    uint32_t first_slot;
    uint32_t n_slots;
} SpxSharedRow;

#endif /* SPX_GPU_FORMAT_H */
