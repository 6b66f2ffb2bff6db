/*
 * nlpkkt_gen.c -- row-sliced generator of the syn-nlpkkt stand-in (bench/test
 * INPUT generation only; nothing here takes part in the SpMV).
 *
 * order n = 2*N^3 + 6*N^2 (N = 240: 27 993 600, the order
 * of SuiteSparse nlpkkt240), KKT layout [H A^T; A D] with 27-point stencils,
 * about 27 nonzeros per row, symmetric with a full diagonal.  The unknowns are
 *
 *     [0, N^3)                 state, one per grid point       (primal)
 *     [N^3, N^3 + 6 N^2)       boundary control, one per face  (primal)
 *     [P, P + N^3)             multipliers, one per grid point (dual), P = N^3 + 6 N^2
 *
 * and the blocks
 *     H  (P x P)     diagonal
 *     A  (N^3 x P)   = [A_y | A_u]: A_y the 27-point stencil of the grid (row s holds
 *                      the columns of the up to 27 neighbours of grid point s, i.e.
 *                      nine runs of three consecutive columns), A_u one entry per
 *                      face: the constraint of the grid point the face belongs to
 *     D  (N^3 x N^3) diagonal
 * so that a state row holds its diagonal and 27 entries of A^T, a multiplier row 27
 * entries of A, its faces and its diagonal: 2*(3N-2)^3 + 12*N^2 + n nonzeros
 * (N = 240: 768 977 264, 27.47 per row; nlpkkt240 itself: 760.6 M, 27.17 per row).
 *
 * Values are a symmetric hash of (min(r,c), max(r,c)) in U(-1,1); the diagonal is
 * 1 + the row's absolute off-diagonal sum.  The same pattern comes out of
 * sparsex_amd/synth.py::syn_nlpkkt (scipy, small N); this file produces it row by
 * row, so that a process can generate just the rows it owns of a matrix that would
 * not fit a scipy pass.  Rounds 1-2 used a different matrix under this name (two
 * fully coupled interleaved fields, 54 nonzeros per row): now tools/synth/kkt2f_gen.c.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

static inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static inline double pair_value(uint64_t seed, int64_t r, int64_t c, int64_t n)
{
    const int64_t a = r < c ? r : c, b = r < c ? c : r;
    const uint64_t h = splitmix64(seed ^ ((uint64_t) a * (uint64_t) n + (uint64_t) b));
    return (double) (h >> 11) * (2.0 / 9007199254740992.0) - 1.0;   /* U(-1, 1) */
}

typedef struct { int64_t N, N2, N3, P, n; } Kkt;

static void kkt_init(Kkt *g, int N)
{
    g->N = N;
    g->N2 = (int64_t) N * N;
    g->N3 = g->N2 * N;
    g->P = g->N3 + 6 * g->N2;
    g->n = g->P + g->N3;
}

static inline int stencil_count(const Kkt *g, int64_t s)
{
    const int64_t N = g->N, z = s / g->N2, y = (s / N) % N, x = s % N;
    const int cz = 1 + (z > 0) + (z < N - 1), cy = 1 + (y > 0) + (y < N - 1),
              cx = 1 + (x > 0) + (x < N - 1);
    return cz * cy * cx;
}

/* the faces grid point s belongs to, ascending; returns their number */
static inline int faces_of(const Kkt *g, int64_t s, int64_t *f)
{
    const int64_t N = g->N, z = s / g->N2, y = (s / N) % N, x = s % N;
    int k = 0;
    if (z == 0) f[k++] = y * N + x;
    if (z == N - 1) f[k++] = g->N2 + y * N + x;
    if (y == 0) f[k++] = 2 * g->N2 + z * N + x;
    if (y == N - 1) f[k++] = 3 * g->N2 + z * N + x;
    if (x == 0) f[k++] = 4 * g->N2 + z * N + y;
    if (x == N - 1) f[k++] = 5 * g->N2 + z * N + y;
    return k;
}

/* the grid point face f belongs to */
static inline int64_t cell_of(const Kkt *g, int64_t f)
{
    const int64_t N = g->N, q = f / g->N2, a = (f % g->N2) / N, b = f % N;
    switch (q) {
    case 0: return a * N + b;
    case 1: return (N - 1) * g->N2 + a * N + b;
    case 2: return a * g->N2 + b;
    case 3: return a * g->N2 + (N - 1) * N + b;
    case 4: return a * g->N2 + b * N;
    default: return a * g->N2 + b * N + (N - 1);
    }
}

int64_t spx_syn_nlpkkt_nrows(int N)
{
    Kkt g;
    kkt_init(&g, N);
    return g.n;
}

/* nnz of every row of the matrix */
void spx_syn_nlpkkt_counts(int N, int32_t *counts)
{
    Kkt g;
    int64_t f[6];
    kkt_init(&g, N);
    for (int64_t s = 0; s < g.N3; ++s) {
        const int c = stencil_count(&g, s);
        counts[s] = 1 + c;
        counts[g.P + s] = c + faces_of(&g, s, f) + 1;
    }
    for (int64_t k = 0; k < 6 * g.N2; ++k) counts[g.N3 + k] = 2;
}

/* the up to 27 neighbours of grid point s, ascending, each added to `base` */
static inline int64_t stencil_cols(const Kkt *g, int64_t s, int64_t base, int32_t *col)
{
    const int64_t N = g->N, z = s / g->N2, y = (s / N) % N, x = s % N;
    int64_t k = 0;
    for (int dz = -1; dz <= 1; ++dz) {
        if (z + dz < 0 || z + dz >= N) continue;
        for (int dy = -1; dy <= 1; ++dy) {
            if (y + dy < 0 || y + dy >= N) continue;
            for (int dx = -1; dx <= 1; ++dx) {
                if (x + dx < 0 || x + dx >= N) continue;
                col[k++] = (int32_t) (base + (z + dz) * g->N2 + (y + dy) * N + (x + dx));
            }
        }
    }
    return k;
}

/* rows [lo, hi): colind/values in CSR order (columns ascending), rowptr relative
 * to the slice (hi-lo+1 entries).  Returns the number of nonzeros written. */
int64_t spx_syn_nlpkkt_rows(int N, int64_t lo, int64_t hi, uint64_t seed, int64_t *rowptr,
                            int32_t *colind, double *values)
{
    Kkt g;
    kkt_init(&g, N);
    int64_t k = 0, f[6];
    rowptr[0] = 0;
    for (int64_t r = lo; r < hi; ++r) {
        const int64_t k0 = k;
        int64_t kd;
        if (r < g.N3) {                       /* state: diagonal, then A_y^T            */
            kd = k;
            colind[k++] = (int32_t) r;
            k += stencil_cols(&g, r, g.P, colind + k);
        } else if (r < g.P) {                 /* control: diagonal, then A_u^T          */
            kd = k;
            colind[k++] = (int32_t) r;
            colind[k++] = (int32_t) (g.P + cell_of(&g, r - g.N3));
        } else {                              /* multiplier: A_y, A_u, diagonal         */
            const int64_t s = r - g.P;
            k += stencil_cols(&g, s, 0, colind + k);
            const int nf = faces_of(&g, s, f);
            for (int i = 0; i < nf; ++i) colind[k++] = (int32_t) (g.N3 + f[i]);
            kd = k;
            colind[k++] = (int32_t) r;
        }
        if (values) {                         /* (NULL: the pattern alone) */
            double sum = 0.0;
            for (int64_t e = k0; e < k; ++e) {
                if (e == kd) continue;
                values[e] = pair_value(seed, r, colind[e], g.n);
                sum += fabs(values[e]);
            }
            values[kd] = sum + 1.0;
        }
        rowptr[r - lo + 1] = k;
    }
    return k;
}

/* The rows rows[0..m) (original numbering, any order) of P A P^T for the permutation perm[old] = new:
 * row i of the result is original row rows[i], its columns renumbered and sorted ascending, its
 * values those of the original matrix.  rowptr has m + 1 entries.  (A process of a multi-GPU run
 * generates just the rows the partition-aware numbering deals it: bench.py --dist-reorder.) */
int64_t spx_syn_nlpkkt_rows_perm(int N, const int64_t *rows, int64_t m, const int32_t *perm, uint64_t seed,
                                 int64_t *rowptr, int32_t *colind, double *values)
{
    int64_t k = 0, rp2[2];
    int32_t c[40];
    double v[40];
    rowptr[0] = 0;
    for (int64_t i = 0; i < m; ++i) {
        const int64_t cnt = spx_syn_nlpkkt_rows(N, rows[i], rows[i] + 1, seed, rp2, c, values ? v : NULL);
        for (int64_t e = 0; e < cnt; ++e) {      /* insertion sort by new column (at most 34 entries) */
            const int32_t cn = perm[c[e]];
            const double vv = values ? v[e] : 0.0;
            int64_t j = k + e;
            while (j > k && colind[j - 1] > cn) {
                colind[j] = colind[j - 1];
                if (values) values[j] = values[j - 1];
                --j;
            }
            colind[j] = cn;
            if (values) values[j] = vv;
        }
        k += cnt;
        rowptr[i + 1] = k;
    }
    return k;
}
