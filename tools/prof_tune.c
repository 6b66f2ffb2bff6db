/* Host-only tune of the nlpkkt stand-in, for a function-level profile of the preprocessor
 * (tools/prof_tune.sh builds it with -pg; no GPU is touched: spx.rt.host_only=true). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <stdint.h>
#include "sparsex/sparsex.h"

void spx_syn_nlpkkt_counts(int N, int32_t *counts);
int64_t spx_syn_nlpkkt_rows(int N, int64_t lo, int64_t hi, uint64_t seed, int64_t *rowptr, int32_t *colind,
                            double *values);

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv)
{
    int64_t edge = argc > 1 ? atoll(argv[1]) : 60;
    const char *threads = argc > 2 ? argv[2] : "1";
    const char *sym = argc > 3 ? argv[3] : "false";
    const int reorder = argc > 4 && strcmp(argv[4], "reorder") == 0;
    int64_t n = 2 * edge * edge * edge + 6 * edge * edge, nnz = 0;
    int32_t *cnt = malloc(n * sizeof *cnt);
    spx_syn_nlpkkt_counts((int) edge, cnt);
    for (int64_t i = 0; i < n; i++) nnz += cnt[i];
    int64_t *rp = malloc((n + 1) * sizeof *rp);
    int32_t *ci = malloc(nnz * sizeof *ci);
    double *va = malloc(nnz * sizeof *va);
    spx_syn_nlpkkt_rows((int) edge, 0, n, 0x5eed, rp, ci, va);
    spx_index_t *rp32 = malloc((n + 1) * sizeof *rp32);
    for (int64_t i = 0; i <= n; i++) rp32[i] = (spx_index_t) rp[i];
    spx_init();
    spx_input_t *in = spx_input_load_csr(rp32, (spx_index_t *) ci, va, (spx_index_t) n, (spx_index_t) n,
                                         SPX_INDEX_ZERO_BASED);
    spx_option_set("spx.rt.host_only", "true");
    spx_option_set("spx.rt.nr_threads", threads);
    spx_option_set("spx.matrix.symmetric", sym);
    spx_log_info_console();
    double t0 = now();
    spx_matrix_t *A = reorder ? spx_mat_tune(in, SPX_MAT_REORDER) : spx_mat_tune(in);
    printf("edge %lld: %lld rows, %lld nonzeros, tune %.2f s (%s threads, symmetric %s%s)\n", (long long) edge,
           (long long) n, (long long) nnz, now() - t0, threads, sym, reorder ? ", RCM reordered" : "");
    spx_mat_destroy(A);
    spx_input_destroy(in);
    return 0;
}
