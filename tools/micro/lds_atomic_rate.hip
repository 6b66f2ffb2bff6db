// Microbenchmark: throughput of ds_add_f64 (no return) into an LDS tile for the
// address patterns the SpMV kernel produces.  One workgroup of 256 threads per
// CU-slot; every lane issues ITER atomics.  Prints lane-atomics per clock per CU.
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/micro/lds_atomic_rate.hip -o gpurun_out/lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ITER = 4096;

template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, const int *pattern, long long *cycles)
{
    __shared__ double tile[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) tile[i] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int idx;
    if (MODE == 0) idx = wave * 64 + lane;                 // distinct, consecutive
    else if (MODE == 1) idx = wave * 64 + (lane >> 3);     // 8 lanes per address
    else if (MODE == 2) idx = wave * 64 + (lane >> 2);     // 4 lanes per address
    else if (MODE == 3) idx = wave;                        // all 64 lanes one address
    else idx = pattern[threadIdx.x] & 4095;                // random
    const double v = 1.0 + lane;
    const long long t0 = clock64();
#pragma unroll 8
    for (int i = 0; i < ITER; ++i) {
        atomicAdd(&tile[(idx + i * (MODE == 4 ? 17 : 0)) & 4095], v);
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (tile[threadIdx.x] == 1.2345) out[0] = tile[threadIdx.x];
}

template <int MODE>
void run(const char *name, int blocks_per_cu)
{
    int ncu = 256;
    double *out; int *pat; long long *cyc;
    hipMalloc(&out, 8); hipMalloc(&pat, 256 * 4); hipMalloc(&cyc, 8 * ncu * blocks_per_cu);
    std::vector<int> hp(256);
    unsigned s = 12345;
    for (int &x : hp) { s = s * 1664525u + 1013904223u; x = (int)(s >> 8); }
    hipMemcpy(pat, hp.data(), 1024, hipMemcpyHostToDevice);
    const int nb = ncu * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<nb, 256>>>(out, pat, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<nb, 256>>>(out, pat, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> hc(nb);
    hipMemcpy(hc.data(), cyc, 8 * nb, hipMemcpyDeviceToHost);
    double avg = 0; for (long long c : hc) avg += c; avg /= nb;
    // clock64 ticks at 100 MHz on this family; use wall time for the rate
    const double lane_atomics_per_cu = 256.0 * ITER * blocks_per_cu;
    const double clk = 2.4e9 * ms * 1e-3;
    printf("%-28s blocks/CU %d: %.3f ms  -> %.2f lane-atomics/clk/CU (at 2.4 GHz), clock64 ticks %.0f\n", name, blocks_per_cu, ms,
           lane_atomics_per_cu / clk, avg);
    hipFree(out); hipFree(pat); hipFree(cyc);
}

int main()
{
    for (int b : {1, 2, 4}) {
        run<0>("distinct consecutive", b);
        run<1>("8 lanes per address", b);
        run<2>("4 lanes per address", b);
        run<3>("64 lanes one address", b);
        run<4>("random", b);
    }
    return 0;
}
