// big_alloc.hpp -- allocator of the preprocessor's multi-gigabyte arrays.
//
// The element arrays of the contract matrix are 25 GB and are first touched by all host threads at
// once; with 4 KB pages that is six million page faults on one address space (they serialise on
// its lock) and as many pages to hand back afterwards -- the time `spx_mat_tune` spent building the
// partitions and releasing them was mostly that.  Blocks of 32 MB and more are therefore mapped on
// their own, aligned to 2 MB and marked for transparent huge pages (a no-op where the system does
// not offer them; SPX_NO_HUGE_PAGES=1 switches the marking off); smaller ones go to malloc as before.
#pragma once

#include <cstddef>
#include <cstdlib>
#include <new>
#include <utility>
#include <sys/mman.h>

namespace spx {

template <class T>
struct BigAlloc {
    typedef T value_type;
    static constexpr size_t kThreshold = (size_t) 32 << 20, kHuge = (size_t) 2 << 20;

    BigAlloc() noexcept {}
    template <class U> BigAlloc(const BigAlloc<U> &) noexcept {}

    static size_t mapped_len(size_t bytes) { return (bytes + kHuge - 1) / kHuge * kHuge; }

    T *allocate(size_t n)
    {
        if (n > (size_t) -1 / sizeof(T)) throw std::bad_alloc();
        const size_t bytes = n * sizeof(T);
        if (bytes < kThreshold) {
            void *p = std::malloc(bytes ? bytes : 1);
            if (!p) throw std::bad_alloc();
            return static_cast<T *>(p);
        }
        const size_t len = mapped_len(bytes);
        char *raw = static_cast<char *>(mmap(nullptr, len + kHuge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
        if (raw == MAP_FAILED) throw std::bad_alloc();
        char *p = reinterpret_cast<char *>(((uintptr_t) raw + kHuge - 1) / kHuge * kHuge);
        if (p > raw) munmap(raw, (size_t) (p - raw));
        if (p + len < raw + len + kHuge) munmap(p + len, (size_t) (raw + len + kHuge - (p + len)));
#ifdef MADV_HUGEPAGE
        // (SPX_NO_HUGE_PAGES=1 in the environment leaves the pages to the system's default: on a host
        // whose memory is fragmented a huge-page fault may wait for compaction)
        static const bool off = std::getenv("SPX_NO_HUGE_PAGES") != nullptr;
        if (!off) madvise(p, len, MADV_HUGEPAGE);
#endif
        return reinterpret_cast<T *>(p);
    }

    void deallocate(T *p, size_t n) noexcept
    {
        const size_t bytes = n * sizeof(T);
        if (bytes < kThreshold) std::free(p);
        else munmap(p, mapped_len(bytes));
    }

    // (elements come into being default-initialised: `v.resize(n)` of doubles or of the plain structs kept
    // here sizes the array without writing to it -- the caller fills it, on several threads where it is
    // large; `v.resize(n, x)` and `v.assign(n, x)` still fill)
    template <class U> void construct(U *p) { ::new (static_cast<void *>(p)) U; }
    template <class U, class A0, class... Args>
    void construct(U *p, A0 &&a0, Args &&...args)
    {
        ::new (static_cast<void *>(p)) U(std::forward<A0>(a0), std::forward<Args>(args)...);
    }

    template <class U> bool operator==(const BigAlloc<U> &) const noexcept { return true; }
    template <class U> bool operator!=(const BigAlloc<U> &) const noexcept { return false; }
};

}  // namespace spx
