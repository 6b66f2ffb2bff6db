/*
 * cpu_baseline.c -- timing harness for the CPU baseline leg of bench.py.
 *
 * TEST INFRASTRUCTURE ONLY (see csx_oracle.c).  Reproduces the execution
 * model of the reference's multithreaded SpMV: persistent worker threads,
 * one partition each, released and joined through a centralized
 * sense-reversing spin barrier around every y <- alpha*A*x
 * (src/internals/CsxKernels.cpp:82-103, src/internals/ThreadPool.cpp:47-116,
 * src/internals/Barrier.cpp:29-60; the caller thread runs partition 0 and
 * zeroes y first, CsxKernels.cpp:93).  The per-partition multiply routine is
 * either the reference's own template code (a function pointer obtained from
 * an oracle/_ref build, kind "reference") or oracle_csx_multiply (kind "port").
 *
 * Symmetric matrices (oracle_time_threads_sym) follow MatVecMult_sym /
 * do_mv_sym_thread (src/internals/CsxKernels.cpp:105-129,
 * src/internals/CsxSpmv.cpp:37-50): every thread but the first multiplies into
 * a full-length local buffer for the rows in front of its own, and a conflict
 * map -- split over the threads by column ranges of equal entry count, as
 * MakeMap does (include/sparsex/internals/CsxBuild.hpp:400-581) -- says which
 * entries are zeroed before and summed into y after the multiply
 * (src/internals/Vector.cpp:213-221, 291-299): four barrier crossings per SpMV.
 */
#define _GNU_SOURCE
#include "csx_oracle.h"

#include <pthread.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef void (*ref_fn_t)(void *spm, void *in, void *out, double scale, void *local);

typedef struct {
    /* "port": m != NULL; "reference": fn + its (opaque) matrix/vector structs */
    const oracle_csx_t *m;
    ref_fn_t fn;
    void *ref_spm, *ref_in, *ref_out;
    const double *x;
    double *y;
    double alpha;
    int cpu;
} slot_t;

typedef struct {              /* map_t, include/sparsex/internals/Map.hpp:23-27 */
    long length;
    unsigned *cpus, *pos;
} cmap_t;

typedef struct {
    int nthreads;
    slot_t *slots;
    /* (a cache line each: the arrivals' read-modify-writes of `count` must not invalidate the
       line the waiters spin on -- with 128 threads on two sockets that is what a barrier costs) */
    _Alignas(64) atomic_int count;
    _Alignas(64) atomic_int sense;
    _Alignas(64) atomic_int stop;
    _Alignas(64) int pad_;
    /* symmetric */
    int symmetric;
    double **locals;          /* [nthreads] raw buffers, locals[0] = y            */
    void **ref_locals;        /* [nthreads] the same as the reference's vector_t  */
    cmap_t *maps;             /* [nthreads]                                       */
} pool_t;

static void barrier_wait(pool_t *p, int *local_sense)
{
    *local_sense = !*local_sense;
    if (atomic_fetch_sub(&p->count, 1) == 1) {
        atomic_store(&p->count, p->nthreads);
        atomic_store(&p->sense, *local_sense);
    } else {
        while (atomic_load_explicit(&p->sense, memory_order_acquire) != *local_sense)
            __builtin_ia32_pause();
    }
}

static void run_slot(slot_t *s)
{
    if (s->fn) s->fn(s->ref_spm, s->ref_in, s->ref_out, s->alpha, NULL);
    else if (s->m) oracle_csx_multiply(s->m, s->x, s->y, s->alpha);
}

/* do_mv_sym_thread */
static void run_slot_sym(pool_t *p, int id, int *sense)
{
    slot_t *s = &p->slots[id];
    const cmap_t *map = &p->maps[id];
    for (long i = 0; i < map->length; i++) p->locals[map->cpus[i]][map->pos[i]] = 0.0;
    barrier_wait(p, sense);
    if (s->fn) s->fn(s->ref_spm, s->ref_in, s->ref_out, s->alpha, p->ref_locals[id]);
    else if (s->m) oracle_csx_sym_multiply(s->m, s->x, s->y, p->locals[id], s->alpha);
    barrier_wait(p, sense);
    for (long i = 0; i < map->length; i++) s->y[map->pos[i]] += p->locals[map->cpus[i]][map->pos[i]];
}

typedef struct { pool_t *p; int id; } warg_t;

static void *worker(void *arg)
{
    warg_t *w = (warg_t *) arg;
    pool_t *p = w->p;
    slot_t *s = &p->slots[w->id];
    if (s->cpu >= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(s->cpu, &set);
        pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
    }
    int sense = 0;
    for (;;) {
        barrier_wait(p, &sense);           /* released by the caller */
        if (atomic_load(&p->stop)) break;
        if (p->symmetric) run_slot_sym(p, w->id, &sense);
        else run_slot(s);
        barrier_wait(p, &sense);           /* joined by the caller */
    }
    return NULL;
}

/* Runs `loops` SpMVs per batch, `batches` batches; returns the median batch
   time in seconds per SpMV.  y (nrows doubles) is zeroed by the caller thread
   before every product, as MatVecMult does. */
static double time_threads(pool_t *preset, int nthreads, const oracle_csx_t *parts, void **ref_fns,
                           void **ref_spms, void *ref_in, void *ref_out, const double *x,
                           double *y, long nrows, double alpha, int loops, int batches,
                           const int *cpus)
{
    pool_t pool = *preset;
    pool.nthreads = nthreads;
    pool.slots = (slot_t *) calloc((size_t) nthreads, sizeof(slot_t));
    atomic_init(&pool.count, nthreads);
    atomic_init(&pool.sense, 0);
    atomic_init(&pool.stop, 0);
    for (int i = 0; i < nthreads; i++) {
        slot_t *s = &pool.slots[i];
        s->m = parts ? &parts[i] : NULL;
        s->fn = ref_fns ? (ref_fn_t) ref_fns[i] : NULL;
        s->ref_spm = ref_spms ? ref_spms[i] : NULL;
        s->ref_in = ref_in;
        s->ref_out = ref_out;
        s->x = x;
        s->y = y;
        s->alpha = alpha;
        s->cpu = cpus ? cpus[i] : -1;
    }
    pthread_t *th = (pthread_t *) calloc((size_t) nthreads, sizeof(pthread_t));
    warg_t *wa = (warg_t *) calloc((size_t) nthreads, sizeof(warg_t));
    for (int i = 1; i < nthreads; i++) {
        wa[i].p = &pool;
        wa[i].id = i;
        pthread_create(&th[i], NULL, worker, &wa[i]);
    }
    if (cpus && cpus[0] >= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(cpus[0], &set);
        pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
    }
    int sense = 0;
    double *times = (double *) calloc((size_t) batches, sizeof(double));
    for (int b = -1; b < batches; b++) {          /* batch -1 warms up */
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int l = 0; l < loops; l++) {
            memset(y, 0, sizeof(double) * (size_t) nrows);
            barrier_wait(&pool, &sense);
            if (pool.symmetric) run_slot_sym(&pool, 0, &sense);
            else run_slot(&pool.slots[0]);
            barrier_wait(&pool, &sense);
        }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (b >= 0)
            times[b] = ((double) (t1.tv_sec - t0.tv_sec) +
                        1e-9 * (double) (t1.tv_nsec - t0.tv_nsec)) / loops;
    }
    atomic_store(&pool.stop, 1);
    barrier_wait(&pool, &sense);
    for (int i = 1; i < nthreads; i++) pthread_join(th[i], NULL);
    /* median */
    for (int i = 0; i < batches; i++)
        for (int j = i + 1; j < batches; j++)
            if (times[j] < times[i]) { double t = times[i]; times[i] = times[j]; times[j] = t; }
    double med = (batches % 2) ? times[batches / 2]
                               : 0.5 * (times[batches / 2 - 1] + times[batches / 2]);
    if (cpus) {
        cpu_set_t all;
        CPU_ZERO(&all);
        for (int c = 0; c < CPU_SETSIZE; c++) CPU_SET(c, &all);
        pthread_setaffinity_np(pthread_self(), sizeof(all), &all);
    }
    free(times); free(wa); free(th); free(pool.slots);
    return med;
}

double oracle_time_threads(int nthreads, const oracle_csx_t *parts, void **ref_fns,
                           void **ref_spms, void *ref_in, void *ref_out, const double *x,
                           double *y, long nrows, double alpha, int loops, int batches,
                           const int *cpus)
{
    pool_t preset;
    memset(&preset, 0, sizeof(preset));
    return time_threads(&preset, nthreads, parts, ref_fns, ref_spms, ref_in, ref_out, x, y, nrows,
                        alpha, loops, batches, cpus);
}

/* Symmetric matrices.  conf_ptr/conf_cols: for every partition k the (ascending)
   columns in front of its first row that it writes (its row of MakeMap's
   initial_map); locals[k] / ref_locals[k]: partition k's full-length buffer
   (k = 0: y itself, as temp[0] = y in MatVecMult_sym). */
double oracle_time_threads_sym(int nthreads, const oracle_csx_t *parts, void **ref_fns,
                               void **ref_spms, void *ref_in, void *ref_out, void **ref_locals,
                               double **locals, const double *x, double *y, long nrows,
                               double alpha, int loops, int batches, const int *cpus,
                               const long *conf_ptr, const int *conf_cols)
{
    pool_t preset;
    memset(&preset, 0, sizeof(preset));
    preset.symmetric = 1;
    preset.locals = locals;
    preset.ref_locals = ref_locals;
    /* MakeMap: count[j] = partitions that write column j; thread i takes the next
       columns until it holds total/(threads left) entries */
    unsigned *count = (unsigned *) calloc((size_t) nrows + 1, sizeof(unsigned));
    long total = 0;
    for (int k = 0; k < nthreads; k++)
        for (long e = conf_ptr[k]; e < conf_ptr[k + 1]; e++) { count[conf_cols[e]]++; total++; }
    long *cursor = (long *) calloc((size_t) nthreads, sizeof(long));
    for (int k = 0; k < nthreads; k++) cursor[k] = conf_ptr[k];
    preset.maps = (cmap_t *) calloc((size_t) nthreads, sizeof(cmap_t));
    long end = 0, left = total;
    for (int i = 0; i < nthreads; i++) {
        const long start = end;
        long take = 0;
        if (i < nthreads - 1) {
            const long limit = left / (nthreads - i);
            while (take < limit && end < nrows) take += count[end++];
        } else {
            end = nrows;
            take = left;
        }
        left -= take;
        cmap_t *m = &preset.maps[i];
        m->length = take;
        m->cpus = (unsigned *) malloc(sizeof(unsigned) * (size_t) (take ? take : 1));
        m->pos = (unsigned *) malloc(sizeof(unsigned) * (size_t) (take ? take : 1));
        long t = 0;
        for (long j = start; j < end; j++)
            for (int k = 0; k < nthreads && count[j]; k++)
                if (cursor[k] < conf_ptr[k + 1] && conf_cols[cursor[k]] == j) {
                    m->cpus[t] = (unsigned) k;
                    m->pos[t++] = (unsigned) j;
                    cursor[k]++;
                }
    }
    const double r = time_threads(&preset, nthreads, parts, ref_fns, ref_spms, ref_in, ref_out, x, y,
                                  nrows, alpha, loops, batches, cpus);
    for (int i = 0; i < nthreads; i++) { free(preset.maps[i].cpus); free(preset.maps[i].pos); }
    free(preset.maps); free(cursor); free(count);
    return r;
}
