/*
 * csx_oracle.c -- CPU restatement of the reference's CSX SpMV path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, imported
 * by or executed from the product (libsparsex.so / sparsex_amd); only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and
 * only as the checker or as the timed CPU baseline.
 *
 * What is restated, and from where (paths inside the reference tree):
 *   - ctl decoding helpers                 include/sparsex/internals/CtlUtil.hpp:46-133
 *   - the general SpMV driver              src/templates/csx_spmv_tmpl.c:66-101
 *     with the hooks CsxJit fills in       include/sparsex/internals/CsxJit.hpp:359-415, 637-672
 *   - per-unit bodies: delta_tmpl.c:20-37, horiz_tmpl.c:20-37, vert_tmpl.c:20-35,
 *     diag_tmpl.c:20-35, rdiag_tmpl.c:20-36, block_row_tmpl.c:20-37,
 *     block_row_one_tmpl.c:20-34, block_col_tmpl.c:20-40, block_col_one_tmpl.c:20-34
 *   - the symmetric driver and bodies      src/templates/csx_sym_spmv_tmpl.c:60-106, *_sym_tmpl.c
 *   - multithreaded dispatch semantics     src/internals/CsxKernels.cpp:35-129,
 *                                          src/internals/CsxSpmv.cpp:28-86
 *   - the CSR check loop and the tolerance test/src/CsxCheck.cpp:28-48,
 *                                          src/internals/Vector.cpp:51-57,396-413
 *
 * The reference JIT-compiles one specialised function per partition; this
 * file interprets the same byte stream with the pattern parameters taken
 * from id_map at run time.  Arithmetic order inside a unit and across units
 * is kept identical to the templates, so for a given stream the result is
 * bit-identical to the reference's generated code (verified against the
 * gcc-compiled templates, see oracle/build_ref.py and tests/).
 *
 * Pinning: fixtures under tests/golden/ were produced by the reference's own
 * templates compiled in place (oracle/_ref) and by the reference's CSR check
 * criterion; see tests/test_oracle_golden.py.
 */
#include "csx_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- ctl helpers ----------------------------------------------------------- */

static inline uint64_t get_fixed(const uint8_t **ctl, int bytes)
{
    uint64_t v = 0;
    memcpy(&v, *ctl, (size_t) bytes);   /* little endian, unaligned */
    *ctl += bytes;
    return v;
}

static inline uint64_t get_varint(const uint8_t **ctl)
{
    uint64_t ret = *(*ctl)++;
    unsigned shift = 7;
    if (ret <= 127) return ret;
    ret -= 128;
    for (;;) {
        uint64_t uc = *(*ctl)++;
        if (uc <= 127) {
            ret += uc << shift;
            break;
        }
        uc -= 128;
        ret += uc << shift;
        shift += 7;
    }
    return ret;
}

#define NR_BIT   0x80u
#define RJMP_BIT 0x40u
#define SLOT_MASK 0x3fu

enum { T_DELTA = 0, T_H = 1, T_V = 2, T_D = 3, T_AD = 4, T_BR1 = 5, T_BR8 = 12,
       T_BC1 = 13, T_BC8 = 20 };

/* ---- general path ------------------------------------------------------------ */

void oracle_csx_multiply(const oracle_csx_t *m, const double *x, double *y, double alpha)
{
    if (m->ctl_size == 0) return;
    const double *v = m->values;
    /* pointer arithmetic on x mirrors the template: a jump may be "negative"
       (wrapped 64-bit value), which wraps around on the address as well */
    const double *x_curr = x;
    double *y_curr = y + m->row_start;
    double yr = 0;
    const uint8_t *ctl = m->ctl;
    const uint8_t *ctl_end = ctl + m->ctl_size;

    do {
        uint8_t flags = *ctl++;
        uint8_t size = *ctl++;
        if (flags & NR_BIT) {
            *y_curr += yr;
            yr = 0;
            if (m->row_jumps && (flags & RJMP_BIT)) y_curr += get_varint(&ctl);
            else y_curr++;
            x_curr = x;
        }
        if (m->full_colind) x_curr = x + get_fixed(&ctl, 4);
        else x_curr = (const double *) ((uintptr_t) x_curr + 8u * get_varint(&ctl));

        long id = m->id_map[flags & SLOT_MASK];
        long type = id / 10000, delta = id % 10000;
        if (type == T_DELTA) {
            int bytes = (int) (delta / 8);
            double r = (*x_curr) * (*v++);
            for (uint8_t i = 1; i < size; i++) {
                x_curr += get_fixed(&ctl, bytes);
                r += (*x_curr) * (*v++);
            }
            yr += r * alpha;
        } else if (type == T_H) {
            double r = 0;
            long i_end = delta * size;
            for (long i = 0; i < i_end; i += delta) r += x_curr[i] * (*v++);
            x_curr += i_end - delta;
            yr += r * alpha;
        } else if (type == T_V) {
            double xr = *x_curr;
            long i_end = delta * size;
            for (long i = 0; i < i_end; i += delta) y_curr[i] += xr * (*v++) * alpha;
        } else if (type == T_D) {
            long i_end = delta * size;
            for (long i = 0; i < i_end; i += delta) y_curr[i] += x_curr[i] * (*v++) * alpha;
        } else if (type == T_AD) {
            const double *xp = x_curr;
            long i_end = delta * size;
            for (long i = 0; i < i_end; i += delta) {
                y_curr[i] += (*xp) * (*v++) * alpha;
                xp -= delta;
            }
        } else if (type == T_BR1) {
            /* 1 x c block: an in-row run */
            double r = 0;
            for (long i = 0; i < delta; i++) r += x_curr[i] * (*v++);
            yr += r * alpha;
        } else if (type > T_BR1 && type <= T_BR8) {
            long r = type - T_BR1 + 1, c = delta;
            for (long i = 0; i < c; i++) {
                double xr = x_curr[i];
                for (long j = 0; j < r; j++) y_curr[j] += xr * (*v++) * alpha;
            }
        } else if (type == T_BC1) {
            /* r x 1 block */
            double xr = *x_curr;
            for (long i = 0; i < delta; i++) y_curr[i] += xr * (*v++) * alpha;
        } else if (type > T_BC1 && type <= T_BC8) {
            long r = delta, c = type - T_BC1 + 1;
            for (long i = 0; i < r; i++) {
                double s = 0;
                for (long j = 0; j < c; j++) s += x_curr[j] * (*v++);
                y_curr[i] += s * alpha;
            }
        } else {
            fprintf(stderr, "[oracle] unknown pattern id %ld\n", id);
            abort();
        }
    } while (ctl < ctl_end);

    *y_curr += yr;
}

/* ---- symmetric path ------------------------------------------------------------- */

void oracle_csx_sym_multiply(const oracle_csx_t *m, const double *x, double *y, double *tmp,
                             double alpha)
{
    const double *v = m->values;
    const double *dv = m->dvalues;
    long x_indx = 0;
    long y_indx = m->row_start;
    const long y_end = (long) m->row_start + m->nrows;
    double yr = 0;
    const uint8_t *ctl = m->ctl;
    const uint8_t *ctl_end = ctl + m->ctl_size;
    double *cur = tmp;

    if (m->ctl_size == 0) goto tail;
    do {
        uint8_t flags = *ctl++;
        uint8_t size = *ctl++;
        if (flags & NR_BIT) {
            y[y_indx] += yr;
            long jmp = 1;
            if (m->row_jumps && (flags & RJMP_BIT)) jmp = (long) get_varint(&ctl);
            for (long i = 0; i < jmp; i++) {
                y[y_indx] += x[y_indx] * (*dv) * alpha;
                y_indx++;
                dv++;
            }
            yr = 0;
            x_indx = 0;
            cur = tmp;
        }
        if (m->full_colind) x_indx = (long) get_fixed(&ctl, 4);
        else x_indx += (long) get_varint(&ctl);
        /* columns inside the partition's own row range go to y directly */
        if (cur != y && x_indx >= m->row_start) cur = y;

        long id = m->id_map[flags & SLOT_MASK];
        long type = id / 10000, delta = id % 10000;
        const double rx0 = x[y_indx];
        if (type == T_DELTA) {
            int bytes = (int) (delta / 8);
            double r = 0, val = *v++;
            r += x[x_indx] * val;
            cur[x_indx] += rx0 * val * alpha;
            for (uint8_t i = 1; i < size; i++) {
                x_indx += (long) get_fixed(&ctl, bytes);
                val = *v++;
                r += x[x_indx] * val;
                cur[x_indx] += rx0 * val * alpha;
            }
            yr += r * alpha;
        } else if (type == T_H) {
            double r = 0;
            long i_end = delta * size;
            for (long i = 0; i < i_end; i += delta) {
                double val = *v++;
                r += x[x_indx + i] * val;
                cur[x_indx + i] += rx0 * val * alpha;
            }
            x_indx += i_end - delta;
            yr += r * alpha;
        } else if (type == T_V) {
            double xv = x[x_indx], ry = 0;
            long i_end = delta * size;
            for (long i = 0; i < i_end; i += delta) {
                double val = *v++;
                y[y_indx + i] += xv * val * alpha;
                ry += x[y_indx + i] * val;
            }
            cur[x_indx] += ry * alpha;
        } else if (type == T_D) {
            long i_end = delta * size;
            for (long i = 0; i < i_end; i += delta) {
                double val = *v++;
                y[y_indx + i] += x[x_indx + i] * val * alpha;
                cur[x_indx + i] += x[y_indx + i] * val * alpha;
            }
        } else if (type == T_AD) {
            /* rdiag_sym_tmpl.c addresses x/cur from (x_indx - i_end) upwards with
               j running i_end .. delta: element i sits at column x_indx - i */
            long i_end = delta * size;
            for (long i = 0; i < i_end; i += delta) {
                double val = *v++;
                y[y_indx + i] += x[x_indx - i] * val * alpha;
                cur[x_indx - i] += x[y_indx + i] * val * alpha;
            }
        } else if (type >= T_BR1 && type <= T_BR8) {
            long r = type - T_BR1 + 1, c = delta;
            for (long i = 0; i < c; i++) {
                double xv = x[x_indx + i], ry = 0;
                for (long j = 0; j < r; j++) {
                    double val = *v++;
                    y[y_indx + j] += xv * val * alpha;
                    ry += x[y_indx + j] * val;
                }
                cur[x_indx + i] += ry * alpha;
            }
        } else if (type >= T_BC1 && type <= T_BC8) {
            long r = delta, c = type - T_BC1 + 1;
            for (long i = 0; i < r; i++) {
                double s = 0, rxv = x[y_indx + i];
                for (long j = 0; j < c; j++) {
                    double val = *v++;
                    s += x[x_indx + j] * val;
                    cur[x_indx + j] += rxv * val * alpha;
                }
                y[y_indx + i] += s * alpha;
            }
        } else {
            fprintf(stderr, "[oracle] unknown pattern id %ld\n", id);
            abort();
        }
    } while (ctl < ctl_end);

    y[y_indx] += yr;
tail:
    for (long i = y_indx; i < y_end; i++) {
        y[i] += x[i] * (*dv) * alpha;
        dv++;
    }
}

/* ---- CSR check loop (test/src/CsxCheck.cpp:28-48) --------------------------------- */

void oracle_csr_spmv(int nrows, const int *rowptr, const int *colind, const double *values,
                     const double *x, double *y)
{
    for (int i = 0; i < nrows; i++) {
        double yr = 0;
        for (int j = rowptr[i]; j < rowptr[i + 1]; j++) yr += values[j] * x[colind[j]];
        y[i] = yr;
    }
}

int oracle_vec_compare(const double *a, const double *b, long n)
{
    /* Vector.cpp:51-57: |(a-b)/a| > 1e-6 is a mismatch (NaN compares equal,
       as in the reference: 0/0 is not > 1e-6) */
    for (long i = 0; i < n; i++)
        if (fabs((a[i] - b[i]) / a[i]) > 1.e-6) return (int) (i + 1);
    return 0;
}

/* ---- whole-matrix products: partitions run like the reference's threads ---------- */

typedef struct {
    const oracle_csx_t *m;
    const double *x;
    double *y;
    double *tmp;
    double alpha;
    int symmetric;
} job_t;

static void *run_job(void *arg)
{
    job_t *j = (job_t *) arg;
    if (j->symmetric) oracle_csx_sym_multiply(j->m, j->x, j->y, j->tmp, j->alpha);
    else oracle_csx_multiply(j->m, j->x, j->y, j->alpha);
    return NULL;
}

void oracle_matvec_mult(const oracle_csx_t *parts, int nparts, int symmetric, long nrows,
                        const double *x, double *y, double alpha, int nthreads,
                        double *scratch)
{
    /* MatVecMult / MatVecMult_sym: y <- 0, every partition accumulates its rows.
       Symmetric: partition p > 0 uses a private full-length buffer which is
       added into y afterwards (the conflict-map reduction of the reference
       visits exactly the entries a partition wrote; adding whole buffers of
       zeros elsewhere gives the same sums in the same per-row order). */
    memset(y, 0, sizeof(double) * (size_t) nrows);
    job_t *jobs = (job_t *) calloc((size_t) nparts, sizeof(job_t));
    for (int p = 0; p < nparts; p++) {
        jobs[p].m = &parts[p];
        jobs[p].x = x;
        jobs[p].y = y;
        jobs[p].alpha = alpha;
        jobs[p].symmetric = symmetric;
        jobs[p].tmp = y;
        if (symmetric && p > 0) {
            jobs[p].tmp = scratch + (size_t) (p - 1) * (size_t) nrows;
            memset(jobs[p].tmp, 0, sizeof(double) * (size_t) nrows);
        }
    }
    if (nthreads <= 1 || nparts <= 1) {
        for (int p = 0; p < nparts; p++) run_job(&jobs[p]);
    } else {
        pthread_t *th = (pthread_t *) calloc((size_t) nparts, sizeof(pthread_t));
        for (int p = 1; p < nparts; p++) pthread_create(&th[p], NULL, run_job, &jobs[p]);
        run_job(&jobs[0]);
        for (int p = 1; p < nparts; p++) pthread_join(th[p], NULL);
        free(th);
    }
    if (symmetric)
        for (int p = 1; p < nparts; p++) {
            const double *t = jobs[p].tmp;
            for (long i = 0; i < nrows; i++) y[i] += t[i];
        }
    free(jobs);
}
