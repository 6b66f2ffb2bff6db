/*
 * sparsex/matvec.h -- matrix and vector routines of the SparseX C API.
 *
 * Every prototype below has the signature of the reference's
 * include/sparsex/matvec.h:39-535 so that existing client code
 * (the examples, test/src/sparsex_test.c, src/bench/SparsexModule.cpp)
 * compiles and links unchanged.  What differs is the machinery behind
 * spx_mat_tune() / spx_matvec_*(): the CSX preprocessor runs on the host and
 * emits a row-block descriptor stream; the multiplication runs as a HIP
 * kernel on an MI355X.  There is no CPU execution path:
 * spx_matvec_*() return SPX_FAILURE when no HIP device is usable.
 */
#ifndef SPARSEX_MATVEC_H
#define SPARSEX_MATVEC_H

#include <sparsex/common.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- input ------------------------------------------------------------ */

/* Wraps user CSR arrays (borrowed, must outlive spx_mat_tune()).  Optional
 * trailing argument: SPX_INDEX_ZERO_BASED (default) or SPX_INDEX_ONE_BASED.
 * ref: src/api/matvec.c:163-215 */
spx_input_t *spx_input_load_csr(const spx_index_t *rowptr,
                                const spx_index_t *colind,
                                const spx_value_t *values,
                                spx_index_t nr_rows,
                                spx_index_t nr_cols, ...);
/* Reads a Matrix Market file with the SparseX extensions.
 * ref: src/api/matvec.c:217-243, include/sparsex/internals/Mmf.hpp:331-478 */
spx_input_t *spx_input_load_mmf(const char *filename);
spx_error_t spx_input_destroy(spx_input_t *input);

/* ---- tuning (CSX preprocessing + upload to HBM) ------------------------ */

/* ref: src/api/matvec.c:259-322.  Optional argument SPX_MAT_REORDER. */
spx_matrix_t *spx_mat_tune(spx_input_t *input, ...);

spx_error_t spx_mat_get_entry(const spx_matrix_t *A, spx_index_t row,
                              spx_index_t column, spx_value_t *value, ...);
spx_error_t spx_mat_set_entry(spx_matrix_t *A, spx_index_t row,
                              spx_index_t column, spx_value_t value, ...);
spx_error_t spx_mat_save(const spx_matrix_t *A, const char *filename);
spx_matrix_t *spx_mat_restore(const char *filename);
spx_index_t spx_mat_get_nrows(const spx_matrix_t *A);
spx_index_t spx_mat_get_ncols(const spx_matrix_t *A);
spx_index_t spx_mat_get_nnz(const spx_matrix_t *A);
spx_partition_t *spx_mat_get_partition(const spx_matrix_t *A);
spx_index_t *spx_partition_get_rs(const spx_partition_t *p);
spx_index_t *spx_partition_get_re(const spx_partition_t *p);
spx_perm_t *spx_mat_get_perm(const spx_matrix_t *A);

/* ---- SpMV --------------------------------------------------------------- */

/* y <- alpha*A*x            ref: src/api/matvec.c:551-584 */
spx_error_t spx_matvec_mult(spx_value_t alpha, const spx_matrix_t *A,
                            const spx_vector_t *x, spx_vector_t *y);
/* y <- alpha*A*x + beta*y   ref: src/api/matvec.c:586-620 */
spx_error_t spx_matvec_kernel(spx_value_t alpha, const spx_matrix_t *A,
                              const spx_vector_t *x, spx_value_t beta,
                              spx_vector_t *y);
/* tunes on first use, then the kernel above   ref: src/api/matvec.c:622-673 */
spx_error_t spx_matvec_kernel_csr(spx_matrix_t **A,
                                  spx_index_t nr_rows, spx_index_t nr_cols,
                                  const spx_index_t *rowptr,
                                  const spx_index_t *colind,
                                  const spx_value_t *values,
                                  spx_value_t alpha, const spx_vector_t *x,
                                  spx_value_t beta, spx_vector_t *y);
spx_error_t spx_mat_destroy(spx_matrix_t *A);

/* ---- partitioning -------------------------------------------------------- */

spx_partition_t *spx_partition_csr(const spx_index_t *rowptr,
                                   spx_index_t nr_rows, size_t nr_threads);
spx_error_t spx_partition_destroy(spx_partition_t *p);

/* ---- options ------------------------------------------------------------- */

/* Mnemonics and defaults: DESIGN.md "Options" (reference: Runtime.cpp:37-95). */
void spx_option_set(const char *option, const char *string);
void spx_options_set_from_env();

/* ---- vectors --------------------------------------------------------------- */

spx_vector_t *spx_vec_create(size_t size, const spx_partition_t *p);
spx_vector_t *spx_vec_create_from_buff(spx_value_t *buff, spx_value_t **tuned,
                                       size_t size, const spx_partition_t *p,
                                       spx_vecmode_t mode);
spx_vector_t *spx_vec_create_random(size_t size, const spx_partition_t *p);
void spx_vec_init(spx_vector_t *v, spx_value_t val);
void spx_vec_init_part(spx_vector_t *v, spx_value_t val, spx_index_t start,
                       spx_index_t end);
void spx_vec_init_rand_range(spx_vector_t *v, spx_value_t max, spx_value_t min);
spx_error_t spx_vec_set_entry(spx_vector_t *v, spx_index_t idx,
                              spx_value_t val, ...);
void spx_vec_scale(spx_vector_t *v1, spx_vector_t *v2, spx_value_t num);
void spx_vec_scale_add(spx_vector_t *v1, spx_vector_t *v2, spx_vector_t *v3,
                       spx_value_t num);
void spx_vec_scale_add_part(spx_vector_t *v1, spx_vector_t *v2,
                            spx_vector_t *v3, spx_value_t num,
                            spx_index_t start, spx_index_t end);
void spx_vec_add(spx_vector_t *v1, spx_vector_t *v2, spx_vector_t *v3);
void spx_vec_add_part(spx_vector_t *v1, spx_vector_t *v2, spx_vector_t *v3,
                      spx_index_t start, spx_index_t end);
void spx_vec_sub(spx_vector_t *v1, spx_vector_t *v2, spx_vector_t *v3);
void spx_vec_sub_part(spx_vector_t *v1, spx_vector_t *v2, spx_vector_t *v3,
                      spx_index_t start, spx_index_t end);
spx_value_t spx_vec_mul(const spx_vector_t *v1, const spx_vector_t *v2);
spx_value_t spx_vec_mul_part(const spx_vector_t *v1, const spx_vector_t *v2,
                             spx_index_t start, spx_index_t end);
spx_error_t spx_vec_reorder(spx_vector_t *v, spx_perm_t *p);
spx_error_t spx_vec_inv_reorder(spx_vector_t *v, spx_perm_t *p);
void spx_vec_copy(const spx_vector_t *v1, spx_vector_t *v2);
int spx_vec_compare(const spx_vector_t *v1, const spx_vector_t *v2);
void spx_vec_print(const spx_vector_t *v);
void spx_vec_destroy(spx_vector_t *v);

#ifdef __cplusplus
}
#endif

#endif /* SPARSEX_MATVEC_H */
