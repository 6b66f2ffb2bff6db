// threads.hpp -- small host-side helpers for work that splits into independent
// pieces (partitions, row ranges, row-blocks): the role the reference's
// per-thread preprocessing plays (include/sparsex/internals/CsxBuild.hpp:290-380),
// without a pool -- tuning is not a hot loop.
#pragma once

#include "common.hpp"

#include <algorithm>
#include <atomic>
#include <exception>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#ifdef __linux__
#include <sched.h>
#endif

namespace spx {

inline unsigned host_threads()
{
    unsigned n = std::thread::hardware_concurrency();
#ifdef __linux__
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
        const int c = CPU_COUNT(&set);
        if (c > 0) n = (unsigned) c;
    }
#endif
    return n ? n : 1u;
}

// fn(i) for i in [0, n) on up to `nthreads` threads, in dynamic order.  The
// first FatalError is rethrown on the caller's thread once all threads joined.
template <typename F>
void parallel_for(size_t n, unsigned nthreads, F fn)
{
    if (nthreads <= 1 || n <= 1) {
        for (size_t i = 0; i < n; ++i) fn(i);
        return;
    }
    std::atomic<size_t> next(0);
    std::mutex mtx;
    std::string err;
    bool failed = false;
    auto work = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= n) return;
            // (whatever a piece throws -- a FatalError, a bad_alloc of a multi-GB vector -- is
            // carried to the caller: an exception leaving a std::thread would end the process)
            auto fail = [&](const std::string &what) {
                std::lock_guard<std::mutex> lk(mtx);
                if (!failed) err = what;
                failed = true;
                next.store(n);
            };
            try {
                fn(i);
            } catch (const FatalError &e) {
                fail(e.what);
            } catch (const std::exception &e) {
                fail(std::string("host worker: ") + e.what());
            } catch (...) {
                fail("host worker: unknown exception");
            }
        }
    };
    const unsigned t = (unsigned) std::min<size_t>(nthreads, n);
    std::vector<std::thread> th;
    th.reserve(t);
    for (unsigned k = 1; k < t; ++k) {
        try {
            th.emplace_back(work);
        } catch (const std::exception &) {
            break;                  // no more threads to be had: the ones running (and the caller) do the work
        }
    }
    work();
    for (auto &x : th) x.join();
    if (failed) throw FatalError(err);
}

}  // namespace spx
