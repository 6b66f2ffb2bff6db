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
 *   spx.gpu.rowblock_elems  target nonzeros per row-block (default 0 = auto:
 *                           nnz/1280 clamped to [1024, 4096]; 8192 beyond 64 M)
 *   spx.gpu.waves           wavefronts per workgroup of the SpMV kernel: 2, 4 or 8;
 *                           0 (default): spx_mat_tune() measures a few launch
 *                           configurations on the device and keeps the fastest
 *   spx.gpu.rowblock_rows   max rows per row-block (default and cap 512)
 *   spx.gpu.stack_segments  "false": one descriptor per CSX unit piece instead
 *                           of merging equal row segments of consecutive rows
 *   spx.gpu.recut_linear    "false": vertical / diagonal / strided units always run one
 *                           nonzero per lane, even where they line up along rows
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
 * summed over processes by the caller (RCCL all-reduce).
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
/* y <- alpha*A*x + beta*y on device vectors (spx_matvec_kernel, matvec.c:586-620) */
spx_error_t spx_hip_matvec_kernel_vec(spx_value_t alpha, const spx_matrix_t *A,
                                      const spx_hip_vec_t *x, spx_value_t beta,
                                      spx_hip_vec_t *y, void *stream);

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
    int32_t pad_;
    double  tune_seconds;    /* preprocessing (mining + encoding)               */
    double  emit_seconds;    /* descriptor stream + upload                      */
} spx_hip_info_t;

spx_error_t spx_hip_mat_info(const spx_matrix_t *A, spx_hip_info_t *info);

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
