/*
 * sparsex_hip.h -- additive extensions of the MI355X build of the SparseX C API.
 *
 * Nothing here changes a signature of <sparsex/sparsex.h>.  The entry points
 * let a caller (a) keep x and y resident in HBM between calls, which is how
 * an iterative solver should drive the library on a GPU, (b) inspect the
 * row-block descriptor stream's size, and (c) export a tuned partition in the
 * reference's own CSX byte format (ctl / values / id_map / rows_info of
 * include/sparsex/internals/Csx.hpp:29-53 in the reference tree) so that the
 * reference's SpMV code, or a parity oracle, can consume it.
 *
 * Extra option mnemonics understood by spx_option_set() in this build:
 *   spx.rt.host_only        "true": spx_mat_tune() only preprocesses (no GPU
 *                           needed); spx_matvec_*() on such a matrix fail
 *   spx.rt.device           HIP device ordinal (default: current device)
 *   spx.rt.gpu_rank/world   this process owns partitions
 *                           [rank*P/world, (rank+1)*P/world), P = spx.rt.nr_threads
 *   spx.rt.row_offset,      the input holds rows [offset, offset + its rows) of a
 *   spx.rt.global_rows      matrix with global_rows rows (0: the input is the matrix)
 *   spx.gpu.rowblock_elems  target nonzeros per row-block (default 0 = auto:
 *                           nnz/1280 clamped to [1024, 4096]; 8192 beyond 64 M)
 *   spx.gpu.waves           wavefronts per workgroup of the SpMV kernel: 2, 4 or 8;
 *                           0 (default): spx_mat_tune() measures a few launch
 *                           configurations on the device and keeps the fastest
 *   spx.gpu.rowblock_rows   max rows per row-block (default 512; up to 2048: planned row-blocks of
 *                           at most 512 rows are then joined up to the target size -- for matrices
 *                           with a few nonzeros per row; measured on syn-webbase: 38.8 us with 512,
 *                           40.9 with 1024, so not the default)
 *   spx.gpu.stack_segments  "false": one descriptor per CSX unit piece instead
 *                           of merging equal row segments of consecutive rows
 *   spx.gpu.recut_linear    "false": vertical / diagonal / strided units always run one
 *                           nonzero per lane, even where they line up along rows
 *   spx.gpu.col_phases      general path: the stream as a sum of K column slices, A = A_0 + A_1 + ..., each a run of
 *                           row-blocks of its own, so that a workgroup only gathers from 1 / K of x (a slice that
 *                           fits the 4 MB of L2 of an XCD).  "c2" | "c4" | "c8": all slices in ONE launch, slice k on
 *                           its own group of 8 / K XCDs, every row-block adding its y tile on top of a beta * y pass
 *                           (global atomics: not with spx.gpu.deterministic); "2" .. "8": the slices launched one
 *                           after the other (slice k > 0 adds to y; measured slower than the plain stream:
 *                           every launch has a ~10 us critical path); "1": off; "auto" (default): for
 *                           leftover-dominated matrices whose x exceeds 6 MB, two and four concurrent slices
 *                           are measured against the plain stream at tune time and the fastest stays
 *                           (syn-webbase: 38.7 us plain, 32.6 us with two slices; profiles/r03/ablation.md)
 *   spx.gpu.band_order      "true": row-blocks are launched strip by strip across the planes of a stencil
 *                           instead of in row order (measured 4-7 % slower on the KKT stand-in: off)
 *   spx.gpu.inline_desc     "false": unit passes whose lanes share one descriptor load it like any other
 *                           (default: it travels in the pass header, SPX_PASSF_INLINE: one dependent round trip
 *                           per pass instead of two; 4-5 % on the KKT stand-in)
 *   spx.gpu.keep_units      "false": ... even those none of whose nonzeros has a neighbour along
 *                           its row (default: such a unit stays one descriptor -- the main diagonal
 *                           of a KKT system -- instead of a leftover nonzero per row)
 *   spx.gpu.wave_tiles      a y tile per wavefront instead of one per workgroup: "true",
 *                           "false", "auto" (default: spx_mat_tune() measures it)
 *   spx.gpu.deterministic   "true": bit-identical repeated products (wave tiles, the
 *                           symmetric tiles' sums through the fixed-order lists)
 *   spx.gpu.sym_spill       symmetric tiles' transposed sums: "lists" (second kernel,
 *                           fixed order), "atomic" (global atomics), "auto" (measured)
 *   spx.gpu.sym_segments    symmetric path: runs of consecutive columns of the lower triangle
 *                           (spx.gpu.sym_segment_min or more of them, default 2; with 3 syn-nlpkkt
 *                           takes 0.880 instead of 0.836 ms) are read once and used for both triangles
 *                           ("true"), their mirror image is stored instead ("false"), or
 *                           "auto" (default): read once where at least half of the triangle
 *                           lies in such runs and the triangle has 16 M nonzeros or more (a
 *                           smaller matrix stays in the Infinity Cache, where reading it
 *                           twice is cheaper than the extra atomics); implies the atomic
 *                           hand-over
 *   spx.gpu.sym_wide_rows   rows of a row-block that holds such segments: consecutive row-blocks
 *                           (512 rows at most each) go side by side into one with a common y
 *                           tile and common slots, so that a column several of them reach is
 *                           handed to y once (default 1024, at most 2048; measured on syn-nlpkkt,
 *                           734 M nonzeros: 512 0.842 ms, 1024 0.826 ms, 2048 0.92 ms)
 *   spx.gpu.sym_pipeline    symmetric path, streams of read-once segments: "auto" (default: measured at tune time) |
 *                           "true" | "false" -- passes whose lanes all belong to one unit carry their geometry in a
 *                           device-side copy of their header, so that values and x are requested in one round trip
 *                           and the passes run as a two-stage pipeline (csx_spmv_sx_kernel; the bench matrix 1.00 ->
 *                           0.91 ms, edges 100-180 10-17 % faster: profiles/r06/NOTES.md)
 *   spx.gpu.sym_pure_passes "true" (default): a long run of equal read-once segments (40 and more) fills passes of its
 *                           own, i.e. passes with ONE descriptor -- what the pipeline above feeds on; "false": passes
 *                           are filled regardless of units, as before round 6
 *   spx.gpu.sym_segment_max widest segment a longer run of columns is cut into (default 8; with 4 every segment could
 *                           ride the pipeline, measured slower: syn-kkt2f 122 -> 135 us, profiles/r06/segment_max_raw.md)
 *   spx.gpu.arena           "true": every array of a tuned matrix in ONE HBM allocation (2 MB-aligned pieces)
 *                           instead of one allocation each (default; the arena was built to test whether
 *                           placement explains the run-to-run spread: it does not, profiles/r04/spread.md)
 *   spx.rt.dist_reorder     the whole matrix given to every process of a multi-GPU job (spx.rt.gpu_world > 1):
 *                           "rcm" | "rcm_owner" renumber the unknowns in front of the nonzero-balanced cut
 *                           (spx_hip_dist_reorder below; spx_mat_get_perm returns the permutation); "none" (default)
 *   spx.rt.dist_chunks      SPX_DIST_OVERLAP: parts the own product is cut into (default 4; 1: no rounds planned --
 *                           on every process or on none: planning the rounds is collective; the counts may differ)
 *   spx.gpu.x_window        "false": leftovers never gather from an LDS window of x
 *   spx.gpu.unit_windows    general path: "auto" (default: measured at tune time) | "true" | "false" -- the columns a
 *                           row-block's unit passes read are staged in LDS once per workgroup and the unit passes run
 *                           as a software pipeline (csx_spmv_xw_kernel); spx.gpu.unit_window_doubles (3072): most
 *                           doubles of x a row-block may stage (row-blocks that need more gather through L2);
 *                           spx.gpu.unit_window_gap (16): column intervals closer than this are staged as one
 *   spx.vec.device          "false" (default) | "true" (also: env SPX_VEC_DEVICE through spx_options_set_from_env):
 *                           vectors the library allocates itself (spx_vec_create, spx_vec_create_random; page-locked)
 *                           keep x's HBM copy between spx_matvec_* calls, reused while no spx_vec_* call has changed
 *                           the vector.  OPT-IN because struct vector_struct is public: a client that writes through
 *                           v->elements must call spx_hip_vec_touch(v) afterwards (a sampled fingerprint of the
 *                           contents is a second net only: a rewritten vector is seen, one poked element may not be).
 *                           Views of user buffers (spx_vec_create_from_buff, both modes) always travel
 *   spx.vec.register        "auto" (default) | "false": the buffer of a view (spx_vec_create_from_buff) of 32 MB or
 *                           more is page-locked where it lies (hipHostRegister) at the view's first spx_matvec_*, and
 *                           released by spx_vec_destroy; it then travels like a vector of the library's own instead
 *                           of through staging memory.  Invisible to the client (it keeps writing through its own
 *                           pointer); a buffer that cannot be locked is staged.  The buffer must outlive the view
 *                           (as src/api/matvec.c:780-815 assumes)
 *   spx.rt.dist_chunks      at most 64 parts (larger values are clamped)
 *   spx.rt.host_parts       spx_matvec_* on host vectors of 32 MB or more: the number of parts the product runs in
 *                           while y travels back (and x up) part by part; "0" (default): 24 where x goes up by need
 *                           (16 on symmetric streams), else 8; at most 64.  Read at every call
 *   spx.gpu.sym_once        "false": symmetric path reads lower triangle and mirror
 *                           image (default: dense 8x8 tiles are read once)
 *   spx.gpu.sym_remine      "false": symmetric path mirrors unit by unit
 *                           instead of re-cutting the upper triangle
 *   spx.rt.keep_encoded     "false": drop the encoded partitions after the
 *                           upload (no get/set entry, no CSX export)
 *   spx.matrix.onedim_blocks  "true" enables br1/bc1 (MatrixOneDimBlocks,
 *                           no mnemonic in the reference: Runtime.cpp:60)
 */
#ifndef SPARSEX_HIP_H
#define SPARSEX_HIP_H

#include <sparsex/sparsex.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- device-resident SpMV -------------------------------------------------
 * Replaces, for GPU-resident vectors, the host-vector entry points
 * spx_matvec_mult / spx_matvec_kernel (reference src/api/matvec.c:551-620).
 * x_dev: ncols doubles, y_dev: nrows doubles, both HBM pointers on the
 * matrix's device.  `stream` is a hipStream_t (NULL = default stream); the
 * call only enqueues work (so loops of these calls can be captured into a
 * hipGraph by stream capture).  With several processes (spx.rt.gpu_world > 1)
 * only the rows of this process' partitions are written on the general path;
 * on the symmetric path y_dev receives this process' partial vector, to be
 * summed over processes by the caller (RCCL all-reduce) -- or use the exchange plan
 * of spx_hip_mat_dist_attach / spx_hip_matvec_dist below, which moves only the
 * entries that have to travel.
 * A matrix handle is single-stream: products of the SAME matrix must not overlap in
 * time (they share its scratch: the partial sums of over-long rows, the spill array
 * of the symmetric tiles, the exchange buffers); different matrices are independent.
 * The calling thread's current HIP device must be the matrix's device (checked).
 * spx_mat_set_entry() on a tuned matrix patches the value where it lives, in HBM, after a
 * hipDeviceSynchronize() -- one per batch of edits, not per entry (the first edit after a product was
 * enqueued waits; it fails, value untouched, while a stream of the device is being captured):
 * products still in flight finish with the old value; a captured
 * graph replayed later reads the new one.
 * Once an exchange plan is attached (spx_hip_mat_dist_attach), the plain entry points above
 * write only the rows this process owns or adds to, [first conflict row, last owned row):
 * the rest of y_dev is left as it is (it is nobody's business any more).
 */
spx_error_t spx_hip_matvec_mult(spx_value_t alpha, const spx_matrix_t *A,
                                const spx_value_t *x_dev, spx_value_t *y_dev,
                                void *stream);
spx_error_t spx_hip_matvec_kernel(spx_value_t alpha, const spx_matrix_t *A,
                                  const spx_value_t *x_dev, spx_value_t beta,
                                  spx_value_t *y_dev, void *stream);

/* ---- device-resident vectors ---------------------------------------------------
 * HBM counterparts of the reference's host vector helpers (spx_vec_init,
 * spx_vec_scale, spx_vec_scale_add, spx_vec_add, spx_vec_sub, spx_vec_mul,
 * spx_vec_copy; reference src/api/matvec.c:838-931, src/internals/
 * Vector.cpp:206-394) so that a CG / GMRES iteration never leaves the GPU.
 * Same argument order and meaning as the host versions; every call enqueues on
 * `stream` (NULL = default stream) and returns immediately, except
 * spx_hip_vec_mul and the download, which synchronise the stream.
 */
typedef struct spx_hip_vec spx_hip_vec_t;

/* spx.vec.device=true only: tells the library that the client wrote to v->elements directly (a vector from
 * spx_vec_create / spx_vec_create_random); the next spx_matvec_* uploads it again.  Harmless otherwise. */
void spx_hip_vec_touch(const spx_vector_t *v);
/* Diagnostic: how the vector's host memory travels.  0: through staging memory (pageable); 1: page-locked memory of the
 * library's own (spx_vec_create, spx_vec_create_random); 2: a client's buffer that the library page-locked in place
 * (spx.vec.register, from the view's first product on); 3: a client's buffer that was page-locked already. */
int spx_hip_vec_page_locked(const spx_vector_t *v);

spx_hip_vec_t *spx_hip_vec_create(size_t size);                 /* zero-filled        */
spx_hip_vec_t *spx_hip_vec_create_from_host(const spx_vector_t *v);
spx_error_t spx_hip_vec_destroy(spx_hip_vec_t *v);
spx_value_t *spx_hip_vec_data(spx_hip_vec_t *v);               /* HBM pointer        */
size_t spx_hip_vec_size(const spx_hip_vec_t *v);
spx_error_t spx_hip_vec_upload(spx_hip_vec_t *dst, const spx_vector_t *src, void *stream);
spx_error_t spx_hip_vec_download(const spx_hip_vec_t *src, spx_vector_t *dst, void *stream);
spx_error_t spx_hip_vec_init(spx_hip_vec_t *v, spx_value_t val, void *stream);
/* v2 <- num * v1 */
spx_error_t spx_hip_vec_scale(const spx_hip_vec_t *v1, spx_hip_vec_t *v2, spx_value_t num,
                              void *stream);
/* v3 <- v1 + num * v2 */
spx_error_t spx_hip_vec_scale_add(const spx_hip_vec_t *v1, const spx_hip_vec_t *v2,
                                  spx_hip_vec_t *v3, spx_value_t num, void *stream);
/* v3 <- v1 + v2,  v3 <- v1 - v2 */
spx_error_t spx_hip_vec_add(const spx_hip_vec_t *v1, const spx_hip_vec_t *v2,
                            spx_hip_vec_t *v3, void *stream);
spx_error_t spx_hip_vec_sub(const spx_hip_vec_t *v1, const spx_hip_vec_t *v2,
                            spx_hip_vec_t *v3, void *stream);
/* *result <- v1 . v2   (deterministic two-stage reduction) */
spx_error_t spx_hip_vec_mul(const spx_hip_vec_t *v1, const spx_hip_vec_t *v2,
                            spx_value_t *result, void *stream);
spx_error_t spx_hip_vec_copy(const spx_hip_vec_t *v1, spx_hip_vec_t *v2, void *stream);
/* Diagnostic: a read stream with the stores of an SpMV in it.  `src` is read in chunks of `chunk_doubles` (a
 * multiple of 2048), one per workgroup, workgroup b on XCD b % 8 as the SpMV kernels' row-blocks; every workgroup
 * then stores `write_doubles` doubles to its stretch of `dst` (dst->size >= chunks * write_doubles).  Timed by
 * the caller; what bench.py prints as roofline.measured_mixed_peak. */
spx_error_t spx_hip_probe_read_write(const spx_hip_vec_t *src, spx_hip_vec_t *dst, size_t chunk_doubles,
                                     size_t write_doubles, void *stream);
/* y <- alpha*A*x + beta*y on device vectors (spx_matvec_kernel, matvec.c:586-620) */
spx_error_t spx_hip_matvec_kernel_vec(spx_value_t alpha, const spx_matrix_t *A,
                                      const spx_hip_vec_t *x, spx_value_t beta,
                                      spx_hip_vec_t *y, void *stream);

/* ---- one process per GPU: row-partitioned matrices -------------------------------
 * The reference partitions the rows over the threads of one process
 * (include/sparsex/internals/SparseInternal.hpp:117-152); its symmetric kernel
 * lets every thread add into rows of the threads in front of it through local
 * buffers, and a conflict map says which entries have to be summed afterwards
 * (src/api/matvec.c:302-318, include/sparsex/internals/CsxBuild.hpp:400-451,
 * src/internals/Vector.cpp:291-299).  Here the "threads" are processes, one per
 * MI355X, and the map drives a point-to-point exchange over xGMI:
 *
 *   - every process tunes the rows it owns, either as a slice of the
 *     partitions of a matrix it was given in full (spx.rt.gpu_rank/gpu_world)
 *     or from just those rows (spx.rt.row_offset / spx.rt.global_rows: the
 *     input of spx_input_load_csr holds rows [offset, offset + nrows) of a
 *     matrix with global_rows rows; symmetric: the full rows, of which the part
 *     on and below the diagonal is kept);
 *   - spx_hip_mat_dist_attach() (collective) learns every process' row range,
 *     hands each process' conflict rows to their owners and builds the lists
 *     for the exchange;
 *   - spx_hip_matvec_dist() multiplies and completes the rows this process
 *     owns: general matrices need nothing from the others; symmetric ones send
 *     the sums they formed for rows in front of their own -- only those entries,
 *     packed -- to the owners, which add them in a fixed order.  With
 *     SPX_DIST_GATHER_Y the finished slices are then handed round so that every
 *     process holds all of y (what a solver needs as the next x).
 *
 * The exchange itself goes through a transport: RCCL over xGMI (built in;
 * librccl is loaded when the transport is created, so single-GPU users never
 * pay for it), or two callbacks of the caller's (the CPU tests run the same
 * plan over gloo).  Vectors are full length (ncols / global rows) on every
 * process, as in the reference, where every thread sees all of x.
 */
typedef struct spx_hip_transport {
    void *ctx;
    int rank, world;
    /* Per-peer exchange of 8-byte words in HOST memory; blocking; every
     * process calls it with matching counts (send_cnt[p] here = recv_cnt[me]
     * on p).  Offsets and counts are in words; entries for `rank` itself are
     * ignored.  Returns 0 on success. */
    int (*exchange_host)(void *ctx, const uint64_t *send, const size_t *send_off,
                         const size_t *send_cnt, uint64_t *recv, const size_t *recv_off,
                         const size_t *recv_cnt);
    /* The same for doubles in DEVICE memory, enqueued on `stream` (hipStream_t). */
    int (*exchange_device)(void *ctx, const spx_value_t *send, const size_t *send_off,
                           const size_t *send_cnt, spx_value_t *recv, const size_t *recv_off,
                           const size_t *recv_cnt, void *stream);
} spx_hip_transport_t;

#define SPX_RCCL_ID_BYTES 128
/* rank 0 creates the id and hands it to the others by whatever means the job
 * has (MPI, a file, torch.distributed); then every process creates its
 * transport -- collective, on the current HIP device. */
spx_error_t spx_hip_rccl_unique_id(void *id);
spx_hip_transport_t *spx_hip_transport_rccl(const void *id, int rank, int world);
void spx_hip_transport_destroy(spx_hip_transport_t *t);
/* The number of ranks the transport's RCCL communicator holds (ncclCommCount), or -1 for a transport that
 * was not made by spx_hip_transport_rccl or where librccl does not export the query. */
int spx_hip_transport_rccl_ranks(const spx_hip_transport_t *t);

/* Collective over the transport's processes; the matrix keeps a copy of *t
 * (the transport must outlive the matrix). */
spx_error_t spx_hip_mat_dist_attach(spx_matrix_t *A, const spx_hip_transport_t *t);

#define SPX_DIST_OWNED_ROWS 0   /* y[row_lo, row_hi) complete on return, the rest unspecified */
#define SPX_DIST_GATHER_Y   1   /* all of y complete on every process */
#define SPX_DIST_HALO_X     2   /* y[row_lo, row_hi) complete, and of the other processes' rows exactly the
                                   entries that THIS process' rows read as x (its halo, see
                                   spx_hip_mat_dist_halo): what an iteration x <- y needs here, nothing
                                   more.  Every owner packs the entries each of the others asked for at
                                   attach time and sends them pairwise; ignored with SPX_DIST_GATHER_Y */
#define SPX_DIST_OVERLAP    4   /* with SPX_DIST_HALO_X on the general path: the own product runs in
                                   spx.rt.dist_chunks launches over consecutive parts of the rows, and the
                                   halo entries of part k travel on a second stream while part k + 1 is
                                   computed; `stream` waits for the last round.  Same result; where the
                                   stream cannot be cut (or on the symmetric path, whose conflict rows are
                                   complete only after the whole product) the plain order is used */
/* y <- alpha*A*x + beta*y over all processes; device pointers of full length,
 * everything enqueued on `stream`.  Collective. */
spx_error_t spx_hip_matvec_dist(spx_value_t alpha, const spx_matrix_t *A,
                                const spx_value_t *x_dev, spx_value_t beta,
                                spx_value_t *y_dev, int flags, void *stream);

/* What the exchange of an attached matrix looks like (arrays owned by the matrix). */
typedef struct {
    int32_t rank, world;
    const spx_index_t *row_lo, *row_hi;     /* [world] rows of every process          */
    int64_t n_send;                         /* conflict rows of this process          */
    const spx_index_t *send_rows;           /* ascending, i.e. grouped by owner       */
    const size_t *send_off, *send_cnt;      /* [world] segment of every owner         */
    int64_t n_recv;                         /* entries this process receives          */
    const size_t *recv_off, *recv_cnt;      /* [world] segment of every sender        */
    int64_t n_fix_rows;                     /* own rows that receive something        */
    const spx_index_t *fix_rows;            /* [n_fix_rows] ascending                 */
    const uint32_t *fix_ptr, *fix_pos;      /* per such row: positions in the receive buffer */
    int32_t any_exchange;                   /* some process sends something           */
} spx_hip_dist_plan_t;
spx_error_t spx_hip_mat_dist_plan(const spx_matrix_t *A, spx_hip_dist_plan_t *plan);

/* Partition-aware numbering for a row-partitioned matrix.  Ranges of rows dealt by nonzeros follow the
 * order in which the application numbers its unknowns; where that order keeps coupled unknowns far
 * apart (a KKT matrix [H A^T; A D]: a range of state rows reads a whole range of multipliers), every
 * process needs a large part of the other processes' vectors.  spx_hip_dist_reorder() computes, from the
 * CSR pattern of the whole (square) matrix, a permutation perm[old] = new (0-based, `nrows` entries,
 * caller's array) after which the ranges couple mostly with themselves:
 *   SPX_DIST_REORDER_RCM        reverse Cuthill-McKee (the order spx_mat_tune(.., SPX_MAT_REORDER) uses)
 *   SPX_DIST_REORDER_RCM_OWNER  that order only decides WHICH of `world` ranges (equal nonzeros) a row
 *                               belongs to; inside a range the rows keep their original order, so the
 *                               substructures of the original numbering (runs, diagonals, blocks) survive
 * `flags`: SPX_DIST_PATTERN_SYMMETRIC promises that the pattern equals its transpose (it is then used
 * where it lies; otherwise A + A^T is built).  The caller applies the permutation to rows, columns and
 * vectors (P A P^T, spx_vec_reorder semantics), e.g. rank 0 computes it once and hands it to the others.
 * With the whole matrix given to every process (spx.rt.gpu_rank / gpu_world), the option
 * spx.rt.dist_reorder = none | rcm | rcm_owner makes spx_mat_tune() do all of this itself
 * (spx_mat_get_perm() returns the permutation).  Reference: its reordering is Rcm.hpp:85-121 on one
 * process; it has no partition-aware form. */
#define SPX_DIST_REORDER_RCM        1
#define SPX_DIST_REORDER_RCM_OWNER  2
#define SPX_DIST_PATTERN_SYMMETRIC  1
spx_error_t spx_hip_dist_reorder(const spx_index_t *rowptr, const spx_index_t *colind, spx_index_t nrows,
                                 int indexing /* SPX_INDEX_ZERO_BASED | SPX_INDEX_ONE_BASED */, int world,
                                 int mode, int flags, spx_index_t *perm);

/* The halo of x of an attached matrix (arrays owned by the matrix): the columns outside its
 * own rows that this process' stream reads (found by walking the stream itself), grouped by
 * owner, and the own rows the other processes asked for, grouped by the process that asked. */
typedef struct {
    int64_t n_recv;                         /* entries this process needs of the others        */
    const spx_index_t *recv_cols;           /* ascending, i.e. grouped by owner                */
    const size_t *recv_off, *recv_cnt;      /* [world] segment of every owner                  */
    int64_t n_send;                         /* own entries the others need (with repetitions)  */
    const spx_index_t *send_rows;           /* grouped by the process that asked, ascending    */
    const size_t *send_off, *send_cnt;      /* [world] segment of every such process           */
} spx_hip_dist_halo_t;
spx_error_t spx_hip_mat_dist_halo(const spx_matrix_t *A, spx_hip_dist_halo_t *halo);

/* y <- alpha*A*x + beta*y in `parts` launches over consecutive parts of the rows (equal work each),
 * one after the other on `stream`: what SPX_DIST_OVERLAP interleaves its rounds with, for a caller that
 * wants to interleave communication of its own (or to measure what cutting the launch costs).  General
 * path, plain streams; where the stream cannot be cut the product runs as one launch.  Returns the number
 * of launches through *launched (may be NULL). */
spx_error_t spx_hip_matvec_parts(spx_value_t alpha, const spx_matrix_t *A, const spx_value_t *x_dev,
                                 spx_value_t beta, spx_value_t *y_dev, int parts, void *stream, int *launched);

/* The rounds of the overlapped step (SPX_DIST_OVERLAP) of an attached matrix: round r moves, per
 * peer, the segment [off, off + cnt) of the halo send list (spx_hip_dist_halo_t::send_rows) out and
 * the segment of the halo receive list (recv_cols) in -- the entries that lie in part r of their
 * owner's rows.  Over all rounds every entry travels exactly once.  0 rounds: none planned
 * (spx.rt.dist_chunks <= 1). */
int spx_hip_mat_dist_rounds(const spx_matrix_t *A);
int spx_hip_mat_dist_parts(const spx_matrix_t *A);     /* launches THIS process' product is cut into (0: not cut) */
spx_error_t spx_hip_mat_dist_round(const spx_matrix_t *A, int round, const size_t **send_off, const size_t **send_cnt,
                                   const size_t **recv_off, const size_t **recv_cnt);

/* ---- introspection ---------------------------------------------------------- */
typedef struct {
    int64_t nnz;             /* logical nonzeros of the whole matrix            */
    int64_t nnz_stored;      /* values held by this process (lower+0 diag on sym) */
    int64_t n_unit_elems;    /* of those, inside substructure units             */
    int64_t n_delta_elems;   /* of those, in delta (leftover) regions           */
    int64_t n_units;         /* unit descriptors                                */
    int64_t n_rowblocks;
    int64_t n_shared_rows;   /* rows split over several row-blocks              */
    int64_t value_bytes;     /* values array incl. alignment padding            */
    int64_t index_bytes;     /* descriptors + start bits + column offsets + ... */
    int32_t nr_partitions;   /* P (all processes)                               */
    int32_t first_partition, last_partition;   /* owned: [first, last)          */
    int32_t row_lo, row_hi;  /* rows owned by this process: [lo, hi)            */
    int32_t symmetric;
    int32_t on_device;       /* 0 for host-only matrices                        */
    int32_t device;
    int32_t waves;           /* wavefronts per workgroup of the SpMV kernel      */
    int32_t sym_tiles;       /* symmetric path: the stream holds dense 8x8 tiles
                                that are read once (SPX_PASS_SYMTILE): 1 = their
                                transposed sums go through the spill lists and a
                                second kernel, 2 = straight into y (global atomics) */
    double  tune_seconds;    /* preprocessing (mining + encoding)               */
    double  emit_seconds;    /* descriptor stream + upload                      */
    int32_t wave_tiles;      /* 1: every wavefront of a workgroup adds into a y tile
                                of its own (summed in wavefront order)          */
    int32_t sym_segments;    /* the stream holds row segments of the lower triangle
                                that are read once and used twice (SPX_PASS_SYMSEG):
                                1 = next to dense tiles, 2 = and no tiles           */
    int32_t quad;            /* reserved (0): round 3's kernel variant with four narrow unit
                                passes side by side per wavefront is in use          */
    int32_t col_slices;      /* general path, spx.gpu.col_phases: K > 1 column slices in one
                                launch (a group of XCDs each), -K: launched in turn, else 1 */
    int32_t unit_windows;    /* general path, spx.gpu.unit_windows: 1 = the product stages the columns
                                its unit passes read in LDS (one set of windows per row-block) and
                                runs the unit passes as a software pipeline (csx_spmv_xw_kernel)   */
    int32_t unit_window_lds; /* ... bytes of LDS per workgroup of that kernel (0: no windows planned) */
    int64_t unit_window_elems;  /* ... nonzeros whose x comes from LDS (of n_unit_elems)            */
    int64_t unit_window_staged; /* ... doubles of x staged per product                              */
    int32_t sym_pipeline;    /* symmetric path, spx.gpu.sym_pipeline: 1 = the read-once passes that carry their
                                geometry in the header run pipelined, x requested with the values
                                (csx_spmv_sx_kernel)                                                  */
    int32_t reserved0;
    int64_t sym_pipeline_elems; /* ... nonzeros in such passes (of the nonzeros in read-once passes)    */
} spx_hip_info_t;

spx_error_t spx_hip_mat_info(const spx_matrix_t *A, spx_hip_info_t *info);

/* The unit windows of x (spx.gpu.unit_windows) planned from the matrix' descriptor stream with the given
 * budget (most doubles of x a row-block may stage in LDS) and gap (column intervals closer than this are
 * staged as one): what csx_spmv_xw_kernel is handed next to the stream -- for inspection and tests; works
 * on host-only matrices.  The arrays belong to the matrix and live until the next call or
 * spx_mat_destroy().  Layouts: sparsex_amd/csrc/xwindows.hpp. */
typedef struct {
    const uint32_t *tab;        /* n_rowblocks x 16 entries of {uint32, uint32}: 2 of pass ranges, 14 windows */
    const uint32_t *xdescs;     /* n_descs x {col0 or LDS offset, bits}                                       */
    const void *passes;         /* n_passes pass headers of 24 bytes (flag 2: the pass reads x from LDS)      */
    size_t n_rowblocks, n_descs, n_passes;
    size_t rowblocks_with_windows, rowblocks_with_units;
    uint64_t staged_doubles;    /* doubles of x staged per product                                            */
    uint64_t unit_elems, unit_elems_lds;   /* nonzeros in unit passes / in unit passes that read LDS          */
    uint32_t lds_doubles;       /* LDS of a launch, doubles: y tile + leftover window + unit windows          */
} spx_hip_xw_plan_t;
spx_error_t spx_hip_mat_unit_windows(spx_matrix_t *A, uint32_t budget, uint32_t gap, spx_hip_xw_plan_t *plan);

/* The device-side pass headers of the pipelined read-once kernel (spx.gpu.sym_pipeline), planned from the
 * matrix' descriptor stream: for inspection and tests; works on host-only matrices.  The arrays belong to the
 * matrix and live until the next call or spx_mat_destroy().  Layout of an SX header: sparsex_amd/csrc/sxplan.hpp. */
typedef struct {
    const void *passes;         /* n_passes pass headers of 24 bytes (flag 4: an SX header)                    */
    const uint32_t *n_sx;       /* per row-block: its passes [0, n_sx) are SX passes                            */
    size_t n_rowblocks, n_passes;
    size_t rowblocks_with_sx;
    uint64_t sym_elems, sx_elems;     /* nonzeros in read-once passes / of those, in SX passes                  */
    uint64_t sym_passes, sx_passes;
} spx_hip_sx_plan_t;
spx_error_t spx_hip_mat_sym_pipeline(spx_matrix_t *A, spx_hip_sx_plan_t *plan);
/* The same for a caller compiled against another revision of this header: at most `size` bytes
 * (the caller's sizeof(spx_hip_info_t)) are written -- the struct only ever grows at its end.
 * spx_hip_mat_info() writes sizeof(spx_hip_info_t) of THIS header; SPX_HIP_ABI_VERSION changes
 * whenever the struct grows (round 3 added quad and col_slices: version 3; round 5 the four
 * unit_window fields: version 4; round 6 the sym_pipeline fields: version 5). */
#define SPX_HIP_ABI_VERSION 5
spx_error_t spx_hip_mat_info_sized(const spx_matrix_t *A, void *info, size_t size);
int spx_hip_abi_version(void);

/* Diagnostic: the number of parts the last spx_matvec_mult / spx_matvec_kernel on HOST vectors ran in (a large y
 * travels back part by part behind the product; 0: the product ran in one piece). */
int spx_hip_mat_host_parts(const spx_matrix_t *A);
/* ... and, where x of that call went up piece by piece in the order the parts needed it (general streams; the part
 * that needs the fewest pieces not yet on the device runs first, so that x on its way up and finished rows of y on
 * their way back share the link), the order the parts ran in: up to `cap` part numbers are written, the number of
 * parts is returned; 0 when x went up as a whole or not at all (resident since the last call). */
int spx_hip_mat_host_order(const spx_matrix_t *A, int32_t *order, int cap);
/* Inspection (host side; works on a matrix tuned with spx.rt.host_only=true): which pieces of x -- of `piece`
 * elements each, at most 64 of them -- every row-block of the stream reads: bit p of mask[i] for the columns
 * [p * piece, (p + 1) * piece), a superset; row0 / n_rows: the row-block's rows.  Any of the arrays may be NULL; up
 * to `cap` entries are written, the number of row-blocks is returned (-1: error).  The plan behind
 * spx_hip_mat_host_order is made of these. */
int64_t spx_hip_mat_x_pieces(spx_matrix_t *A, size_t piece, uint64_t *mask, uint32_t *row0, uint32_t *n_rows, size_t cap);

/* ---- export in the reference's CSX layout --------------------------------------
 * `part` is a global partition number owned by this process.  The arrays
 * stay owned by the matrix and live until spx_mat_destroy().
 */
typedef struct {
    const spx_value_t *values;   /* nnz values in unit order                    */
    const uint8_t *ctl;          /* ctl byte stream                              */
    int64_t ctl_size;
    spx_index_t nnz, ncols, nrows, row_start;
    int32_t row_jumps;           /* 1 if any unit carries a row jump             */
    int32_t full_colind;         /* 1: 32-bit absolute columns, 0: varint jumps  */
    long id_map[64];             /* slot -> pattern id, -1 terminated            */
    const spx_index_t *rows_info;/* nrows x {rowptr, valptr, span}               */
    const spx_value_t *dvalues;  /* symmetric: nrows diagonal values, else NULL  */
} spx_csx_export_t;

spx_error_t spx_hip_mat_export_csx(const spx_matrix_t *A, int part,
                                   spx_csx_export_t *out);

/* One record per unit of a partition after preprocessing, in row-major anchor
 * order: {type, delta, size, row, col} (1-based coordinates inside the
 * partition; type 0 = single leftover nonzero).  Returns the number of
 * records; fills at most `cap` of them. */
typedef struct {
    int32_t type, delta, size, row, col;
} spx_unit_record_t;

int64_t spx_hip_mat_export_units(const spx_matrix_t *A, int part,
                                 spx_unit_record_t *recs, int64_t cap);

/* The preprocessing log of the last spx_mat_tune() of this matrix (statistics
 * per round, chosen encodings); NUL-terminated, owned by the matrix. */
const char *spx_hip_mat_tune_log(const spx_matrix_t *A);

/* The coordinate maps between iteration orders used by the preprocessor
 * (reference include/sparsex/internals/Xform.hpp:37-248): transforms the
 * 1-based (*row, *col) from order `from` to order `to` (EncType numbering:
 * 1 h, 2 v, 3 d, 4 ad, 5..12 br1..8, 13..20 bc1..8).  Exposed for tests. */
void spx_hip_xform(int from, int to, spx_index_t *row, spx_index_t *col,
                   spx_index_t nr_rows, spx_index_t nr_cols);

/* Restores every option to its default (the reference keeps options in a
 * process-wide singleton with no reset; tests need one). */
void spx_hip_options_reset(void);

#ifdef __cplusplus
}
#endif

#endif /* SPARSEX_HIP_H */
