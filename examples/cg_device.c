/*
 * cg_device.c -- conjugate gradients on a 2-D Laplacian with everything resident
 * in HBM: the matrix tuned by spx_mat_tune(), the vectors as spx_hip_vec_t, one
 * scalar read back per dot product.  Plain C against the C ABI of libsparsex.so:
 *
 *   gcc examples/cg_device.c -Iinclude -Lsparsex_amd/lib -lsparsex \
 *       -Wl,-rpath,$PWD/sparsex_amd/lib -lm -o cg_device && ./cg_device 300
 *
 * The host-vector calls of the reference API (spx_matvec_mult, spx_vec_*) would
 * run the same loop with two PCIe copies per product; the spx_hip_* calls below
 * are their device-resident counterparts with the same argument order.
 */
#include <sparsex/sparsex.h>
#include <sparsex_hip.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

int main(int argc, char **argv)
{
    const int g = argc > 1 ? atoi(argv[1]) : 200;      /* g x g grid */
    const int n = g * g;
    spx_index_t *rowptr = malloc((n + 1) * sizeof(*rowptr));
    spx_index_t *colind = malloc((size_t) 5 * n * sizeof(*colind));
    spx_value_t *values = malloc((size_t) 5 * n * sizeof(*values));
    int nnz = 0;
    for (int i = 0; i < g; i++)
        for (int j = 0; j < g; j++) {
            const int r = i * g + j;
            rowptr[r] = nnz;
            if (i > 0) { colind[nnz] = r - g; values[nnz++] = -1.0; }
            if (j > 0) { colind[nnz] = r - 1; values[nnz++] = -1.0; }
            colind[nnz] = r; values[nnz++] = 4.0;
            if (j < g - 1) { colind[nnz] = r + 1; values[nnz++] = -1.0; }
            if (i < g - 1) { colind[nnz] = r + g; values[nnz++] = -1.0; }
        }
    rowptr[n] = nnz;

    spx_init();
    spx_option_set("spx.matrix.symmetric", "true");     /* stored once, used twice */
    spx_input_t *in = spx_input_load_csr(rowptr, colind, values, n, n, SPX_INDEX_ZERO_BASED);
    spx_matrix_t *A = spx_mat_tune(in);
    if (!A) return 1;

    /* b = A * ones, so the solution is the vector of ones */
    spx_hip_vec_t *x = spx_hip_vec_create(n), *b = spx_hip_vec_create(n);
    spx_hip_vec_t *r = spx_hip_vec_create(n), *p = spx_hip_vec_create(n), *ap = spx_hip_vec_create(n);
    spx_hip_vec_init(p, 1.0, NULL);
    spx_hip_matvec_kernel_vec(1.0, A, p, 0.0, b, NULL);
    spx_hip_vec_copy(b, r, NULL);                        /* x0 = 0  =>  r0 = b */
    spx_hip_vec_copy(r, p, NULL);
    double rr, rr0, pap;
    spx_hip_vec_mul(r, r, &rr, NULL);
    rr0 = rr;
    int it = 0;
    while (rr > 1e-20 * rr0 && it < 10 * g) {
        spx_hip_matvec_kernel_vec(1.0, A, p, 0.0, ap, NULL);       /* ap = A p        */
        spx_hip_vec_mul(p, ap, &pap, NULL);
        const double alpha = rr / pap;
        spx_hip_vec_scale_add(x, p, x, alpha, NULL);                 /* x += alpha p    */
        spx_hip_vec_scale_add(r, ap, r, -alpha, NULL);               /* r -= alpha ap   */
        double rr_new;
        spx_hip_vec_mul(r, r, &rr_new, NULL);
        spx_hip_vec_scale_add(r, p, p, rr_new / rr, NULL);           /* p = r + beta p  */
        rr = rr_new;
        it++;
    }

    spx_value_t *xh = malloc(n * sizeof(*xh));
    spx_vector_t *xv = spx_vec_create_from_buff(xh, NULL, n, NULL, SPX_VEC_AS_IS);
    spx_hip_vec_download(x, xv, NULL);
    double err = 0.0;
    for (int i = 0; i < n; i++) err = fmax(err, fabs(xh[i] - 1.0));
    printf("n = %d, nnz = %d: %d CG iterations, |r|/|b| = %.3e, max |x - 1| = %.3e\n", n, nnz, it,
           sqrt(rr / rr0), err);

    spx_vec_destroy(xv);
    spx_hip_vec_destroy(x); spx_hip_vec_destroy(b); spx_hip_vec_destroy(r);
    spx_hip_vec_destroy(p); spx_hip_vec_destroy(ap);
    spx_mat_destroy(A);
    spx_input_destroy(in);
    free(rowptr); free(colind); free(values); free(xh);
    return err < 1e-6 ? 0 : 2;
}
