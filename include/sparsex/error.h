/*
 * sparsex/error.h -- error codes and the replaceable error handler.
 *
 * Same codes, macros and handler signature as the reference
 * (include/sparsex/error.h:28-145): routines return SPX_SUCCESS/SPX_FAILURE or
 * an SPX_INVALID_* handle after calling the current handler.
 */
#ifndef SPARSEX_ERROR_H
#define SPARSEX_ERROR_H

#include <stdio.h>
#include <stdlib.h>
#include <errno.h>
#include <stdarg.h>
#include <string.h>
#include <assert.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPX_FAILURE             -1
#define SPX_SUCCESS             0

/* library errors */
#define SPX_ERR_MIN_VALUE       1
#define SPX_ERR_ARG_INVALID     2
#define SPX_ERR_FILE            3
#define SPX_ERR_INPUT_MAT       4
#define SPX_ERR_TUNED_MAT       5
#define SPX_ERR_VEC             6
#define SPX_ERR_PART            7
#define SPX_ERR_PERM            8
#define SPX_ERR_DIM             9
#define SPX_ERR_VEC_DIM         10
#define SPX_ERR_ENTRY_NOT_FOUND 11
#define SPX_OUT_OF_BOUNDS       12
/* system errors (the default handler exits on these) */
#define SPX_ERR_SYSTEM          15
#define SPX_ERR_FILE_OPEN       16
#define SPX_ERR_FILE_READ       17
#define SPX_ERR_FILE_WRITE      18
#define SPX_ERR_MEM_ALLOC       19
#define SPX_ERR_MEM_FREE        20
#define SPX_ERR_MAX_VALUE       21
/* warnings */
#define SPX_WARN_CSXFILE         22
#define SPX_WARN_TUNING_OPT      23
#define SPX_WARN_RUNTIME_OPT     24
#define SPX_WARN_REORDER         25
#define SPX_WARN_ENTRY_NOT_SET   26
#define SPX_WARN_MAX_VALUE       27

typedef int spx_error_t;

typedef void (*spx_errhandler_t)(spx_error_t, const char *, unsigned long,
                                 const char *, const char *, ...);

#define SETERROR_0(code) \
    spx_err_get_handler()(code, __FILE__, __LINE__, __func__, NULL)
#define SETERROR_1(code, message) \
    spx_err_get_handler()(code, __FILE__, __LINE__, __func__, message)
#define SETWARNING(code) \
    spx_err_get_handler()(code, __FILE__, __LINE__, __func__, NULL)

/* default handler: prints "<message> ["file":line:func()]" to stderr */
void err_handle(spx_error_t code, const char *sourcefile, unsigned long lineno,
                const char *function, const char *fmt, ...);
spx_errhandler_t spx_err_get_handler();
void spx_err_set_handler(spx_errhandler_t new_handler);

#ifdef __cplusplus
}
#endif

#endif /* SPARSEX_ERROR_H */
