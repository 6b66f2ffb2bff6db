/*
 * CsxCheck.hpp -- what the reference's test client (test/src/sparsex_test.c)
 * includes for its result check.  The reference's own header pulls in its
 * C++ internals (Boost); this one only declares the C function the client
 * calls.  Own text; implementation in check_result.c.
 */
#ifndef SPX_REF_CLIENT_CSXCHECK_HPP
#define SPX_REF_CLIENT_CSXCHECK_HPP

#include <sparsex/sparsex.h>

#ifdef __cplusplus
extern "C" {
#endif
/* y ?= alpha * A * x with A re-read from `matrix_file` as CSR; exits with
 * status 1 on a mismatch (relative 1e-6, the reference's criterion), prints
 * "Check Passed" otherwise. */
void check_result(spx_vector_t *result, double alpha, spx_vector_t *x, char *matrix_file);
#ifdef __cplusplus
}
#endif

#endif
