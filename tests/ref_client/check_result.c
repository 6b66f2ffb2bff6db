/*
 * check_result.c -- the result check of the reference's test client, restated
 * for files in the reference's sorted coordinate format ("rows cols nnz" and
 * one "row col value" line per nonzero, 1-based, optional %-comment lines):
 * a serial product from the file's entries, compared with the reference's
 * criterion |a - b| / |a| <= 1e-6 (test/src/CsxCheck.cpp:28-48,
 * src/internals/Vector.cpp:51-57, :396-413).
 */
#include "CsxCheck.hpp"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

void check_result(spx_vector_t *result, double alpha, spx_vector_t *x, char *matrix_file)
{
    FILE *f = fopen(matrix_file, "r");
    if (!f) { fprintf(stderr, "check: cannot open %s\n", matrix_file); exit(1); }
    char line[512];
    long nrows = 0, ncols = 0;
    double nnz = 0;
    while (fgets(line, sizeof(line), f))
        if (line[0] != '%' && sscanf(line, "%ld %ld %lf", &nrows, &ncols, &nnz) == 3) break;
    if (nrows <= 0 || (size_t) nrows != result->size || (size_t) ncols != x->size) {
        fprintf(stderr, "check: dimensions do not match the vectors\n");
        exit(1);
    }
    double *y = calloc((size_t) nrows, sizeof(double));
    long r, c;
    double v;
    long seen = 0;
    while (fgets(line, sizeof(line), f)) {
        if (sscanf(line, "%ld %ld %lf", &r, &c, &v) != 3) continue;
        y[r - 1] += v * x->elements[c - 1];
        ++seen;
    }
    fclose(f);
    printf("Checking... ");
    for (long i = 0; i < nrows; i++) {
        const double a = alpha * y[i], b = result->elements[i];
        if (fabs(a - b) > 1e-6 * fabs(a) && fabs(a - b) > 1e-300) {
            printf("element %ld differs: %.17g != %.17g (%ld entries read)\n", i, a, b, seen);
            exit(1);
        }
    }
    printf("Check Passed\n");
    free(y);
}
