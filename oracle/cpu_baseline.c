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

typedef struct {
    int nthreads;
    slot_t *slots;
    atomic_int count;
    atomic_int sense;
    atomic_int stop;
} pool_t;

static void barrier_wait(pool_t *p, int *local_sense)
{
    *local_sense = !*local_sense;
    if (atomic_fetch_sub(&p->count, 1) == 1) {
        atomic_store(&p->count, p->nthreads);
        atomic_store(&p->sense, *local_sense);
    } else {
        while (atomic_load(&p->sense) != *local_sense)
            __builtin_ia32_pause();
    }
}

static void run_slot(slot_t *s)
{
    if (s->fn) s->fn(s->ref_spm, s->ref_in, s->ref_out, s->alpha, NULL);
    else if (s->m) oracle_csx_multiply(s->m, s->x, s->y, s->alpha);
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
        run_slot(s);
        barrier_wait(p, &sense);           /* joined by the caller */
    }
    return NULL;
}

/* Runs `loops` SpMVs per batch, `batches` batches; returns the median batch
   time in seconds per SpMV.  y (nrows doubles) is zeroed by the caller thread
   before every product, as MatVecMult does. */
double oracle_time_threads(int nthreads, const oracle_csx_t *parts, void **ref_fns,
                           void **ref_spms, void *ref_in, void *ref_out, const double *x,
                           double *y, long nrows, double alpha, int loops, int batches,
                           const int *cpus)
{
    pool_t pool;
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
            run_slot(&pool.slots[0]);
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
