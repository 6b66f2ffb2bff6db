// Microbenchmark: what a read-only stream reaches on this box (the roof the SpMV kernel's
// value stream is measured against): every lane keeps U 16-byte loads in flight over a
// buffer far beyond the Infinity Cache.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/stream_read.hip -o gpurun_out/stream_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int U>
__global__ __launch_bounds__(256) void read_kernel(const double2 *p, size_t n2, double *out)
{
    const size_t stride = (size_t) gridDim.x * 256;
    size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    double acc = 0.0;
    for (; i + (U - 1) * stride < n2; i += U * stride) {
        double2 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = p[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; ++k) acc += v[k].x + v[k].y;
    }
    if (acc == 1.2345) out[0] = acc;
}

template <int U>
static void run(const double2 *p, size_t n2, double *out, unsigned blocks)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    read_kernel<U><<<blocks, 256>>>(p, n2, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) read_kernel<U><<<blocks, 256>>>(p, n2, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("U=%d blocks=%5u: %.1f GB/s\n", U, blocks, 10.0 * n2 * 16 / (ms * 1e6));
}

int main()
{
    const size_t bytes = (size_t) 6 << 30;
    double2 *p; double *out;
    if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) return 1;
    hipMemset(p, 0, bytes);
    const size_t n2 = bytes / 16;
    for (unsigned blocks : {1024u, 2048u, 4096u, 8192u, 16384u}) {
        run<1>(p, n2, out, blocks);
        run<4>(p, n2, out, blocks);
        run<8>(p, n2, out, blocks);
    }
    return 0;
}
