/*
 * dist_spmv.c -- a row-partitioned symmetric SpMV over several MI355X, one process per
 * GPU, through the C ABI only (include/sparsex/sparsex.h + include/sparsex_hip.h).
 *
 *   ./dist_spmv <world> <rank> <id-file> [n]
 *
 * Every process builds ONLY the rows it owns of a banded symmetric n x n matrix (full rows,
 * global column numbers), tunes them on its GPU (device = rank), joins the exchange and
 * multiplies.  Rank 0 creates the RCCL unique id and writes it to <id-file>; the others wait
 * for the file -- any other way of handing 128 bytes round (MPI_Bcast, a socket) does as well.
 * With world = 1 it runs on a single GPU (nothing to exchange).
 *
 * gcc examples/dist_spmv.c -Iinclude -Lsparsex_amd/lib -lsparsex -lm -o dist_spmv
 */
#include <sparsex/sparsex.h>
#include <sparsex_hip.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define BAND 4          /* a(i, i +- k) = -1 / k for k = 1..BAND, a(i, i) = 5 */

static double entry(long i, long j) { return i == j ? 5.0 : -1.0 / (double) labs(i - j); }

int main(int argc, char **argv)
{
    if (argc < 4) {
        fprintf(stderr, "usage: %s <world> <rank> <id-file> [n]\n", argv[0]);
        return 2;
    }
    const int world = atoi(argv[1]), rank = atoi(argv[2]);
    const char *idfile = argv[3];
    const long n = argc > 4 ? atol(argv[4]) : 200000;
    const long lo = n * rank / world, hi = n * (rank + 1) / world, rows = hi - lo;

    /* the rows [lo, hi) as CSR: full rows, zero-based global columns */
    spx_index_t *rowptr = malloc(sizeof(spx_index_t) * (size_t) (rows + 1));
    spx_index_t *colind = malloc(sizeof(spx_index_t) * (size_t) rows * (2 * BAND + 1));
    spx_value_t *values = malloc(sizeof(spx_value_t) * (size_t) rows * (2 * BAND + 1));
    long nnz = 0;
    for (long i = lo; i < hi; i++) {
        rowptr[i - lo] = (spx_index_t) nnz;
        for (long j = i - BAND; j <= i + BAND; j++)
            if (j >= 0 && j < n) {
                colind[nnz] = (spx_index_t) j;
                values[nnz++] = entry(i, j);
            }
    }
    rowptr[rows] = (spx_index_t) nnz;

    spx_init();
    char buf[32];
    snprintf(buf, sizeof buf, "%d", rank);
    spx_option_set("spx.rt.device", buf);                    /* one GPU per process */
    spx_option_set("spx.matrix.symmetric", "true");
    if (world > 1) {
        snprintf(buf, sizeof buf, "%ld", lo);
        spx_option_set("spx.rt.row_offset", buf);
        snprintf(buf, sizeof buf, "%ld", n);
        spx_option_set("spx.rt.global_rows", buf);
    }
    spx_input_t *in = spx_input_load_csr(rowptr, colind, values, (spx_index_t) rows, (spx_index_t) n,
                                         SPX_INDEX_ZERO_BASED);
    spx_matrix_t *A = spx_mat_tune(in);
    if (!A) return 1;

    /* the transport: rank 0 makes the id, everybody joins */
    char id[SPX_RCCL_ID_BYTES];
    if (rank == 0) {
        if (spx_hip_rccl_unique_id(id) != SPX_SUCCESS) return 1;
        char tmp[512];
        snprintf(tmp, sizeof tmp, "%s.tmp", idfile);
        FILE *f = fopen(tmp, "wb");
        if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) return 1;
        fclose(f);
        rename(tmp, idfile);
    } else {
        FILE *f = NULL;
        for (int tries = 0; tries < 6000 && !(f = fopen(idfile, "rb")); tries++) usleep(10000);
        if (!f || fread(id, 1, sizeof id, f) != sizeof id) return 1;
        fclose(f);
    }
    spx_hip_transport_t *t = spx_hip_transport_rccl(id, rank, world);
    if (!t || spx_hip_mat_dist_attach(A, t) != SPX_SUCCESS) return 1;

    /* x = 1: y_i = sum of row i, the same on every rank once the slices went round */
    spx_hip_vec_t *x = spx_hip_vec_create((size_t) n), *y = spx_hip_vec_create((size_t) n);
    spx_hip_vec_init(x, 1.0, NULL);
    if (spx_hip_matvec_dist(1.0, A, spx_hip_vec_data(x), 0.0, spx_hip_vec_data(y), SPX_DIST_GATHER_Y, NULL) !=
        SPX_SUCCESS)
        return 1;
    spx_partition_t *part = spx_mat_get_partition(A);
    spx_vector_t *yh = spx_vec_create((size_t) n, part);
    spx_hip_vec_download(y, yh, NULL);
    double err = 0.0;
    for (long i = 0; i < n; i++) {
        double want = 0.0;
        for (long j = i - BAND; j <= i + BAND; j++)
            if (j >= 0 && j < n) want += entry(i, j);
        err = fmax(err, fabs(yh->elements[i] - want));
    }
    printf("rank %d of %d: rows [%ld, %ld), %ld nonzeros, max |y - exact| = %.3e\n", rank, world, lo, hi, nnz, err);

    /* an iteration x <- y does not need all of y on every rank: SPX_DIST_HALO_X brings, besides the own rows,
       exactly the entries of the others' rows that THIS rank's rows read (here: BAND entries on either side) */
    spx_hip_dist_halo_t halo;
    if (spx_hip_mat_dist_halo(A, &halo) != SPX_SUCCESS) return 1;
    spx_hip_vec_init(y, -7.0, NULL);
    if (spx_hip_matvec_dist(1.0, A, spx_hip_vec_data(x), 0.0, spx_hip_vec_data(y), SPX_DIST_HALO_X | SPX_DIST_OVERLAP,
                            NULL) != SPX_SUCCESS)
        return 1;
    spx_hip_vec_download(y, yh, NULL);
    double err_h = 0.0;
    for (long k = -1; k < halo.n_recv + rows; k++) {
        const long i = k < 0 ? lo : (k < halo.n_recv ? (long) halo.recv_cols[k] : lo + (k - halo.n_recv));
        double want = 0.0;
        for (long j = i - BAND; j <= i + BAND; j++)
            if (j >= 0 && j < n) want += entry(i, j);
        err_h = fmax(err_h, fabs(yh->elements[i] - want));
    }
    printf("rank %d of %d: halo of x: %ld entries received, %ld sent; max |y - exact| on own rows + halo = %.3e\n", rank,
           world, (long) halo.n_recv, (long) halo.n_send, err_h);
    err = fmax(err, err_h);
    /* the figures the predicted step is made of (tests/test_dist_plan_volumes.py holds them):
       bytes received per step, the busiest peer (one xGMI link carries that), rounds of the overlapped step */
    {
        size_t busiest = 0;
        for (int q = 0; q < world; q++) {
            if (halo.recv_cnt[q] > busiest) busiest = halo.recv_cnt[q];
            if (halo.send_cnt[q] > busiest) busiest = halo.send_cnt[q];
        }
        printf("rank %d of %d: per step %.3f MB received (a whole slice handed round: %.3f MB), busiest peer %.3f MB, "
               "%d rounds behind %d launches\n", rank, world, 8e-6 * (double) halo.n_recv, 8e-6 * (double) (n - rows),
               8e-6 * (double) busiest, spx_hip_mat_dist_rounds(A), spx_hip_mat_dist_parts(A));
    }

    spx_hip_vec_destroy(x);
    spx_hip_vec_destroy(y);
    spx_vec_destroy(yh);
    spx_partition_destroy(part);
    spx_mat_destroy(A);
    spx_hip_transport_destroy(t);
    spx_input_destroy(in);
    free(rowptr); free(colind); free(values);
    return err < 1e-9 ? 0 : 1;
}
