/*
 * kkt2f_gen.c -- row-sliced generator of syn-kkt2f (bench/test INPUT generation
 * only; nothing here takes part in the SpMV).
 *
 * Rounds 1-2 used this matrix as their nlpkkt stand-in; it is kept for comparison
 * under a name of its own because it is NOT the matrix SURVEY section 8(d)
 * specifies: two interleaved fields that are fully coupled at every point of the
 * 27-point stencil give 54 nonzeros per row in runs of six consecutive columns,
 * twice the density of nlpkkt240.  (tools/synth/nlpkkt_gen.c is the stand-in.)
 *
 * Same pattern as sparsex_amd/synth.py::syn_kkt2f (two interleaved fields on an
 * N^3 grid with 27-point stencils plus 6*N^2 constraint rows), but produced row
 * by row, so that a process can generate just the rows it owns.  Values are a
 * symmetric hash of (min(r,c), max(r,c)) in U(-1,1); the diagonal is 1 + the
 * row's absolute off-diagonal sum, like the scipy generator's.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

static inline uint64_t splitmix64_2f(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static inline double pair_value_2f(uint64_t seed, int64_t r, int64_t c, int64_t n)
{
    const int64_t a = r < c ? r : c, b = r < c ? c : r;
    const uint64_t h = splitmix64_2f(seed ^ ((uint64_t) a * (uint64_t) n + (uint64_t) b));
    return (double) (h >> 11) * (2.0 / 9007199254740992.0) - 1.0;   /* U(-1, 1) */
}

typedef struct {
    int64_t N, n1, n2, n;
    int64_t *ext_ptr;      /* per H row: extra columns (constraint couplings)  */
    int32_t *ext_col;
} Gen;

static void gen_init(Gen *g, int N)
{
    g->N = N;
    g->n1 = 2 * (int64_t) N * N * N;
    g->n2 = N >= 3 ? 6 * (int64_t) N * N : 0;
    g->n = g->n1 + g->n2;
    g->ext_ptr = (int64_t *) calloc((size_t) g->n1 + 2, sizeof(int64_t));
    g->ext_col = (int32_t *) malloc(sizeof(int32_t) * (size_t) (2 * g->n2 + 1));
    for (int64_t f = 0; f < g->n2; ++f) {
        const int64_t t = (f * 7919) % g->n1, t1 = (t + 1) % g->n1;
        ++g->ext_ptr[t + 2];
        ++g->ext_ptr[t1 + 2];
    }
    for (int64_t r = 0; r < g->n1; ++r) g->ext_ptr[r + 2] += g->ext_ptr[r + 1];
    /* ext_ptr[r+1] is now the fill cursor of row r */
    for (int64_t f = 0; f < g->n2; ++f) {
        const int64_t t = (f * 7919) % g->n1, t1 = (t + 1) % g->n1;
        g->ext_col[g->ext_ptr[t + 1]++] = (int32_t) (g->n1 + f);
        g->ext_col[g->ext_ptr[t1 + 1]++] = (int32_t) (g->n1 + f);
    }
}

static void gen_free(Gen *g)
{
    free(g->ext_ptr);
    free(g->ext_col);
}

static inline int stencil_count_2f(int64_t N, int64_t s)
{
    const int64_t z = s / (N * N), y = (s / N) % N, x = s % N;
    const int cz = 1 + (z > 0) + (z < N - 1), cy = 1 + (y > 0) + (y < N - 1),
              cx = 1 + (x > 0) + (x < N - 1);
    return cz * cy * cx;
}

/* nnz of every row of the matrix */
void spx_syn_kkt2f_counts(int N, int32_t *counts)
{
    Gen g;
    gen_init(&g, N);
    for (int64_t r = 0; r < g.n1; ++r)
        counts[r] = 2 * stencil_count_2f(g.N, r >> 1) + (int32_t) (g.ext_ptr[r + 1] - g.ext_ptr[r]);
    for (int64_t f = 0; f < g.n2; ++f) counts[g.n1 + f] = 3;
    gen_free(&g);
}

int64_t spx_syn_kkt2f_nrows(int N)
{
    return 2 * (int64_t) N * N * N + (N >= 3 ? 6 * (int64_t) N * N : 0);
}

/* rows [lo, hi): colind/values in CSR order (columns ascending), rowptr relative
 * to the slice (hi-lo+1 entries).  Returns the number of nonzeros written. */
int64_t spx_syn_kkt2f_rows(int N, int64_t lo, int64_t hi, uint64_t seed, int64_t *rowptr,
                            int32_t *colind, double *values)
{
    Gen g;
    gen_init(&g, N);
    int64_t k = 0;
    rowptr[0] = 0;
    for (int64_t r = lo; r < hi; ++r) {
        const int64_t k0 = k;
        int64_t kd = -1;
        if (r < g.n1) {
            const int64_t s = r >> 1;
            const int64_t z = s / (g.N * g.N), y = (s / g.N) % g.N, x = s % g.N;
            for (int dz = -1; dz <= 1; ++dz) {
                if (z + dz < 0 || z + dz >= g.N) continue;
                for (int dy = -1; dy <= 1; ++dy) {
                    if (y + dy < 0 || y + dy >= g.N) continue;
                    for (int dx = -1; dx <= 1; ++dx) {
                        if (x + dx < 0 || x + dx >= g.N) continue;
                        const int64_t d = (z + dz) * g.N * g.N + (y + dy) * g.N + (x + dx);
                        colind[k++] = (int32_t) (2 * d);
                        colind[k++] = (int32_t) (2 * d + 1);
                    }
                }
            }
            for (int64_t e = g.ext_ptr[r]; e < g.ext_ptr[r + 1]; ++e) colind[k++] = g.ext_col[e];
        } else {
            const int64_t f = r - g.n1;
            const int64_t t = (f * 7919) % g.n1, t1 = (t + 1) % g.n1;
            colind[k++] = (int32_t) (t < t1 ? t : t1);
            colind[k++] = (int32_t) (t < t1 ? t1 : t);
            colind[k++] = (int32_t) r;
        }
        double sum = 0.0;
        for (int64_t e = k0; e < k; ++e) {
            if (colind[e] == r) { kd = e; continue; }
            values[e] = pair_value_2f(seed, r, colind[e], g.n);
            sum += fabs(values[e]);
        }
        values[kd] = sum + 1.0;
        rowptr[r - lo + 1] = k;
    }
    gen_free(&g);
    return k;
}
