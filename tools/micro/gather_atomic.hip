// Microbenchmarks behind two design decisions of the symmetric and the leftover paths.
//
//  gather <MB> <n>   n random 8-byte gathers from an array of <MB> megabytes.  With the array far
//                    beyond L2 + Infinity Cache every gather is a miss of known size, which
//                    calibrates what rocprofv3's FETCH_SIZE counts per scattered 8-byte read
//                    (run under `rocprofv3 --pmc FETCH_SIZE`); with 8 MB it is syn-webbase's x.
//  atomic <rows> <groups> <per>
//                    groups x per global_atomic_add_f64 (no return) into a vector of <rows>
//                    doubles: `per` consecutive doubles at a random multiple of `per` -- the
//                    transposed sums of a symmetric 8x8 tile handed straight to y (per = 8), or
//                    single scattered adds (per = 1).  Compared with plain coalesced stores of the
//                    same count (what the spill array costs today).
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/micro/gather_atomic.hip -o gpurun_out/gather_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void gather_kernel(const double *x, const uint32_t *idx, double *out, size_t n)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0.0;
    for (size_t k = i; k < n; k += (size_t) gridDim.x * blockDim.x) s += x[idx[k]];
    if (s == 1.2345) out[0] = s;
}

__global__ void atomic_kernel(double *y, const uint32_t *base, size_t groups, int per)
{
    const size_t t = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    const size_t g = t / per;
    if (g < groups) atomicAdd(&y[base[g] + (uint32_t) (t % per)], 1.0 + (double) (t & 7));
}

__global__ void store_kernel(double *spill, size_t n)
{
    const size_t t = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) spill[t] = 1.0 + (double) (t & 7);
}

static float time_ms(void (*fn)(void *), void *arg, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    fn(arg);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) fn(arg);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

struct G { const double *x; const uint32_t *idx; double *out; size_t n; };
struct A { double *y; const uint32_t *base; size_t groups; int per; double *spill; };

int main(int argc, char **argv)
{
    if (argc < 2) return 1;
    unsigned long long s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    if (!strcmp(argv[1], "gather")) {
        const size_t mb = strtoull(argv[2], 0, 10), n = strtoull(argv[3], 0, 10);
        const size_t elems = mb * (1u << 20) / 8;
        double *x, *out; uint32_t *idx;
        CK(hipMalloc(&x, elems * 8)); CK(hipMemset(x, 0, elems * 8));
        CK(hipMalloc(&out, 8)); CK(hipMalloc(&idx, n * 4));
        std::vector<uint32_t> h(n);
        for (auto &v : h) v = (uint32_t) (rnd() % elems);
        CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
        G g{x, idx, out, n};
        const float ms = time_ms([](void *p) { G *g = (G *) p; gather_kernel<<<2048, 256>>>(g->x, g->idx, g->out, g->n); }, &g, 20);
        printf("gather: %zu MB array, %zu gathers: %.2f us  -> %.1f gathers/ns; index stream %.1f MB\n", mb, n, ms * 1e3,
               n / (ms * 1e6), n * 4 / 1e6);
    } else {
        const size_t rows = strtoull(argv[2], 0, 10), groups = strtoull(argv[3], 0, 10);
        const int per = atoi(argv[4]);
        double *y, *spill; uint32_t *base;
        CK(hipMalloc(&y, rows * 8)); CK(hipMemset(y, 0, rows * 8));
        CK(hipMalloc(&spill, groups * per * 8));
        CK(hipMalloc(&base, groups * 4));
        std::vector<uint32_t> h(groups);
        for (auto &v : h) v = (uint32_t) ((rnd() % (rows / per)) * per);
        CK(hipMemcpy(base, h.data(), groups * 4, hipMemcpyHostToDevice));
        A a{y, base, groups, per, spill};
        const unsigned blocks = (unsigned) ((groups * per + 255) / 256);
        const float t_at = time_ms([](void *p) { A *a = (A *) p; atomic_kernel<<<(unsigned) ((a->groups * a->per + 255) / 256), 256>>>(a->y, a->base, a->groups, a->per); }, &a, 20);
        const float t_st = time_ms([](void *p) { A *a = (A *) p; store_kernel<<<(unsigned) ((a->groups * a->per + 255) / 256), 256>>>(a->spill, a->groups * a->per); }, &a, 20);
        printf("atomic: %zu rows, %zu groups x %d: global_atomic_add_f64 %.2f us (%.1f lane-atomics/ns); plain coalesced stores of the same count %.2f us; %u blocks\n",
               rows, groups, per, t_at * 1e3, groups * per / (t_at * 1e6), t_st * 1e3, blocks);
    }
    return 0;
}
