/* PCIe-inclusive rate of spx_matvec_mult from C: vectors created by the library
 * (page-locked, copied directly) vs views of malloc'ed user buffers (staged).
 * usage: host_api_bench <file.mtx> ; build: gcc tools/host_api_bench.c -Iinclude -Lsparsex_amd/lib -lsparsex */
#include <sparsex/sparsex.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv)
{
    if (argc < 2) return 1;
    spx_init();
    spx_input_t *in = spx_input_load_mmf(argv[1]);
    spx_matrix_t *A = spx_mat_tune(in);
    if (!A) return 1;
    spx_partition_t *p = spx_mat_get_partition(A);
    const size_t n = spx_mat_get_nrows(A), m = spx_mat_get_ncols(A);
    spx_vector_t *x = spx_vec_create_random(m, p), *y = spx_vec_create(n, p);
    double *xb = malloc(m * sizeof(double)), *yb = malloc(n * sizeof(double));
    for (size_t i = 0; i < m; i++) xb[i] = 0.01 * (double) (i % 17);
    spx_vector_t *xu = spx_vec_create_from_buff(xb, NULL, m, p, SPX_VEC_AS_IS);
    spx_vector_t *yu = spx_vec_create_from_buff(yb, NULL, n, p, SPX_VEC_AS_IS);
    const int loops = 300;
    for (int w = 0; w < 2; w++) {
        spx_vector_t *a = w ? xu : x, *b = w ? yu : y;
        for (int i = 0; i < 20; i++) spx_matvec_mult(0.5, A, a, b);
        double t0 = now();
        for (int i = 0; i < loops; i++) spx_matvec_mult(0.5, A, a, b);
        double t = (now() - t0) / loops;
        printf("%s: %.1f us per spx_matvec_mult (%.1f GFLOP/s)\n", w ? "user buffers (staged)   " : "library vectors (pinned)",
               1e6 * t, 2.0 * spx_mat_get_nnz(A) / t / 1e9);
    }
    return 0;
}
