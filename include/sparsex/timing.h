/*
 * sparsex/timing.h -- header-only stopwatch used by SparseX client code.
 * Same API as the reference's include/sparsex/timing.h:25-85 (spx_timer_t with
 * clear/start/pause/get_secs), reimplemented on clock_gettime(CLOCK_MONOTONIC).
 */
#ifndef SPARSEX_TIMING_H
#define SPARSEX_TIMING_H

#include <time.h>

struct timer {
    double elapsed;        /* accumulated seconds */
    struct timespec mark;  /* last start() */
};
typedef struct timer spx_timer_t;

static inline void spx_timer_clear(spx_timer_t *t)
{
    t->elapsed = 0.0;
    t->mark.tv_sec = 0;
    t->mark.tv_nsec = 0;
}

static inline void spx_timer_start(spx_timer_t *t)
{
    clock_gettime(CLOCK_MONOTONIC, &t->mark);
}

static inline void spx_timer_pause(spx_timer_t *t)
{
    struct timespec now;
    clock_gettime(CLOCK_MONOTONIC, &now);
    t->elapsed += (double)(now.tv_sec - t->mark.tv_sec) +
                  1e-9 * (double)(now.tv_nsec - t->mark.tv_nsec);
}

static inline double spx_timer_get_secs(spx_timer_t *t)
{
    return t->elapsed;
}

#endif /* SPARSEX_TIMING_H */
