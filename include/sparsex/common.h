/*
 * sparsex/common.h -- handle types, option constants, logging switches.
 *
 * Drop-in for the reference's include/sparsex/common.h:24-286.  The vector
 * structure is public ABI there (Vector.hpp:30-35; check_vec_dim() below reads
 * x->size from client code), so its layout is kept field for field.
 */
#ifndef SPARSEX_COMMON_H
#define SPARSEX_COMMON_H

#include <sparsex/error.h>
#include <sparsex/types.h>
#include <stdlib.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Public layout (reference: include/sparsex/internals/Vector.hpp:30-35). */
struct vector_struct {
    spx_value_t *elements;   /* host buffer, `size` doubles                  */
    size_t size;
    int alloc_type;          /* who owns `elements` (library / user buffer)  */
    int vec_mode;            /* SPX_VEC_AS_IS / SPX_VEC_TUNE / invalid       */
};
typedef struct vector_struct vector_t;

typedef struct matrix spx_matrix_t;
typedef struct input spx_input_t;
typedef struct vector_struct spx_vector_t;
typedef struct partition spx_partition_t;
typedef spx_index_t spx_perm_t;
typedef int spx_option_t;
typedef unsigned int spx_vecmode_t;

#define SPX_INVALID_INPUT   ((spx_input_t *) NULL)
#define SPX_INVALID_MAT     ((spx_matrix_t *) NULL)
#define SPX_INVALID_VEC     ((spx_vector_t *) NULL)
#define SPX_INVALID_PART    ((spx_partition_t *) NULL)
#define SPX_INVALID_PERM    ((spx_perm_t *) NULL)

#define SPX_MAT_REORDER         42
#define SPX_VEC_AS_IS           43
#define SPX_VEC_TUNE            44
#define SPX_INDEX_ZERO_BASED    45
#define SPX_INDEX_ONE_BASED     46

static inline int check_indexing(spx_option_t base)
{
    return (base == SPX_INDEX_ZERO_BASED || base == SPX_INDEX_ONE_BASED);
}

static inline int check_vecmode(spx_vecmode_t mode)
{
    return (mode == SPX_VEC_AS_IS || mode == SPX_VEC_TUNE);
}

static inline int check_mat_dim(spx_index_t dim)
{
    return (dim >= 0);
}

static inline int check_vec_dim(const spx_vector_t *x, unsigned long dim)
{
    return (x->size == dim);
}

/* Logging switches (reference: common.h:171-260).  This build has a single
 * stderr/file sink with a level threshold. */
void spx_log_disable_all();
void spx_log_error_console();
void spx_log_warning_console();
void spx_log_info_console();
void spx_log_verbose_console();
void spx_log_debug_console();
void spx_log_error_file();
void spx_log_warning_file();
void spx_log_info_file();
void spx_log_verbose_file();
void spx_log_debug_file();
void spx_log_all_console();
void spx_log_all_file(const char *file);
void spx_log_set_file(const char *file);

void spx_init();
void spx_finalize();

#define spx_malloc(type, size) \
    (type *) malloc_internal(size, __FILE__, __LINE__, __func__)
void *malloc_internal(size_t x, const char *sourcefile, unsigned long lineno,
                      const char *function);

#define spx_free(object) \
    free_internal(object, __FILE__, __LINE__, __func__)
void free_internal(void *ptr, const char *sourcefile, unsigned long lineno,
                   const char *function);

#ifdef __cplusplus
}
#endif

#endif /* SPARSEX_COMMON_H */
