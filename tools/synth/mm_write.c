/*
 * mm_write.c -- writes triplets as a standard Matrix Market coordinate file
 * (bench/test INPUT generation only).  The caller passes the entries in the
 * order they are to appear; SuiteSparse files are column-ordered, and a
 * symmetric file holds the lower triangle (row >= column) only.
 */
#include <stdint.h>
#include <stdio.h>

int64_t spx_mm_write(const char *path, int symmetric, int64_t nrows, int64_t ncols, int64_t nnz,
                     const int32_t *rows, const int32_t *cols, const double *vals)
{
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    static char buf[1 << 22];
    setvbuf(f, buf, _IOFBF, sizeof(buf));
    fprintf(f, "%%%%MatrixMarket matrix coordinate real %s\n", symmetric ? "symmetric" : "general");
    fprintf(f, "%% written by tools/mm_write.py (sparsex_amd synthetic stand-in)\n");
    fprintf(f, "%lld %lld %lld\n", (long long) nrows, (long long) ncols, (long long) nnz);
    for (int64_t k = 0; k < nnz; ++k)
        fprintf(f, "%d %d %.17g\n", rows[k] + 1, cols[k] + 1, vals[k]);
    if (fclose(f) != 0) return -1;
    return nnz;
}
