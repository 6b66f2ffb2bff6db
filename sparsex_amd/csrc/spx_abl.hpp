// spx_abl.hpp -- ablation switches of the symmetric kernels, all in one place.
//
// The product build defines none of the SPX_ABL_* macros: every switch below is then a constant
// `false` and the branches it guards fold away (tests/test_build_hygiene.py checks both that the
// default build leaves them off and that every variant still compiles).  An experiment build
// (tools/build_variant.sh <name> "-DSPX_ABL_...") leaves one cost out at a time; its results are WRONG
// on purpose and tools/abl.py labels its rows INVALID.  What each variant measured: profiles/r05/ablation.md
// section 2, profiles/r06/sym_writes_raw.md.
#pragma once

namespace spx {
namespace abl {

#ifdef SPX_ABL_SYM_STREAM            /* the stream, the row sums and the init pass only */
#define SPX_ABL_SYM_NOSLOTADD
#define SPX_ABL_SYM_NOX
#define SPX_ABL_SYM_NOHANDOVER
#endif
#ifdef SPX_ABL_SYM_NOWRITES          /* nothing of the product reaches y: no init pass, no own rows, no hand-over */
#define SPX_ABL_SYM_NOINIT
#define SPX_ABL_SYM_NOOWN
#define SPX_ABL_SYM_NOHANDOVER
#endif

#ifdef SPX_ABL_SYM_NOSLOTADD         /* no LDS adds of the transposed products */
constexpr bool sym_no_slot_add = true;
#else
constexpr bool sym_no_slot_add = false;
#endif
#ifdef SPX_ABL_SYM_ONEADD            /* one LDS add per lane instead of W */
constexpr bool sym_one_add = true;
#else
constexpr bool sym_one_add = false;
#endif
#ifdef SPX_ABL_SYM_NOX               /* every x load from one cached line */
constexpr bool sym_no_x = true;
#else
constexpr bool sym_no_x = false;
#endif
#ifdef SPX_ABL_SYM_NOHANDOVER        /* the transposed sums are not added to y */
constexpr bool sym_no_handover = true;
#else
constexpr bool sym_no_handover = false;
#endif
#ifdef SPX_ABL_SYM_NOOWN             /* the own rows are neither stored nor added */
constexpr bool sym_no_own = true;
#else
constexpr bool sym_no_own = false;
#endif
#ifdef SPX_ABL_SYM_NOINIT            /* csx_sym_init_kernel is not launched */
constexpr bool sym_no_init = true;
#else
constexpr bool sym_no_init = false;
#endif
#ifdef SPX_ABL_SYM_NOPRIVATE         /* SPX_RB_PRIVATE ignored: own rows added on top of a full init pass */
constexpr bool sym_no_private = true;
#else
constexpr bool sym_no_private = false;
#endif

#ifdef SPX_ABL_SYM_NOTILERUN         /* tile passes fetch their descriptors themselves (the form before round 6) */
constexpr bool sym_no_tile_run = true;
#else
constexpr bool sym_no_tile_run = false;
#endif

#ifdef SPX_ABL_SYM_NOMIXED           /* the pipelined kernel skips the read-once passes that hold several units */
constexpr bool sym_no_mixed = true;
#else
constexpr bool sym_no_mixed = false;
#endif

}  // namespace abl
}  // namespace spx
