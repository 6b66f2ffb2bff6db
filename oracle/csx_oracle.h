/*
 * csx_oracle.h -- interface of the CPU parity oracle (test infrastructure;
 * see the header of csx_oracle.c for scope and references).
 */
#ifndef CSX_ORACLE_H
#define CSX_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* one partition in the reference's CSX layout (Csx.hpp:37-53) */
typedef struct {
    const double *values;
    const uint8_t *ctl;
    int64_t ctl_size;
    int32_t nnz, ncols, nrows, row_start;
    int32_t row_jumps;     /* stream uses row-jump varints                     */
    int32_t full_colind;   /* 32-bit absolute columns instead of varint jumps  */
    long id_map[64];       /* slot -> type*10000+delta, -1 terminated          */
    const double *dvalues; /* symmetric: diagonal of the partition's rows      */
} oracle_csx_t;

/* y[row_start..] += alpha * A_part * x      (spm_csx_multiply) */
void oracle_csx_multiply(const oracle_csx_t *m, const double *x, double *y, double alpha);
/* symmetric partition: lower triangle + diagonal; columns left of row_start
   accumulate into tmp (spm_csx_sym_multiply) */
void oracle_csx_sym_multiply(const oracle_csx_t *m, const double *x, double *y, double *tmp,
                             double alpha);
/* plain CSR product used by the reference's tests as ground truth */
void oracle_csr_spmv(int nrows, const int *rowptr, const int *colind, const double *values,
                     const double *x, double *y);
/* 0 when equal within the reference's relative 1e-6, else 1-based index */
int oracle_vec_compare(const double *a, const double *b, long n);
/* y <- alpha*A*x over all partitions; scratch: (nparts-1)*nrows doubles (sym) */
void oracle_matvec_mult(const oracle_csx_t *parts, int nparts, int symmetric, long nrows,
                        const double *x, double *y, double alpha, int nthreads,
                        double *scratch);

#ifdef __cplusplus
}
#endif

#endif /* CSX_ORACLE_H */
