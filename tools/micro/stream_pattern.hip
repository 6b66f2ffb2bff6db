// Microbenchmark: what a read-only stream reaches when it is read the way the SpMV kernels read the
// values of a matrix -- every workgroup its own contiguous chunk (a row-block's 64 KB), workgroup b on
// XCD b % 8, each XCD walking its own contiguous eighth of the buffer -- against the grid-stride sweep
// of tools/micro/stream_read.hip in which the whole chip moves through the buffer together.
//   mode 0  grid-stride sweep (the "read roof" probe)
//   mode 1  chunk per workgroup, XCD x owns the x-th eighth of the chunks (the SpMV launch order)
//   mode 2  chunk per workgroup, chunks dealt round robin over the XCDs (chunk = blockIdx)
// LDS bytes per workgroup (argv[2]) bound the workgroups per CU like the kernels' tiles and windows do.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/stream_pattern.hip -o gpurun_out/stream_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int U, int MODE>
__global__ __launch_bounds__(256) void pattern_kernel(const double2 *p, size_t n_chunks, unsigned chunk16, double *out)
{
    extern __shared__ double lds[];
    double acc = 0.0;
    if (MODE == 0) {
        const size_t n2 = n_chunks * chunk16, stride = (size_t) gridDim.x * 256;
        for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n2; i += U * stride) {
            double2 v[U];
#pragma unroll
            for (int k = 0; k < U; ++k) v[k] = p[i + k * stride];
#pragma unroll
            for (int k = 0; k < U; ++k) acc += v[k].x + v[k].y;
        }
    } else {
        size_t chunk;
        if (MODE == 1) {
            const size_t per = (n_chunks + 7) / 8;
            chunk = (size_t) (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
            if ((blockIdx.x >> 3) >= per) return;
        } else {
            chunk = blockIdx.x;
        }
        if (chunk >= n_chunks) return;
        const double2 *q = p + chunk * chunk16;
        // wave w takes pieces w, w + 4, ... of 64 x 16 bytes, U of them in flight
        for (unsigned i = threadIdx.x; i + (U - 1) * 256u < chunk16; i += U * 256u) {
            double2 v[U];
#pragma unroll
            for (int k = 0; k < U; ++k) v[k] = q[i + k * 256u];
#pragma unroll
            for (int k = 0; k < U; ++k) acc += v[k].x + v[k].y;
        }
    }
    if (acc == 1.2345) out[0] = acc + lds[0];
}

template <int U, int MODE>
static void run(const double2 *p, size_t n_chunks, unsigned chunk16, double *out, unsigned blocks, size_t lds)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&pattern_kernel<U, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    pattern_kernel<U, MODE><<<blocks, 256, lds>>>(p, n_chunks, chunk16, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) pattern_kernel<U, MODE><<<blocks, 256, lds>>>(p, n_chunks, chunk16, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d U=%d chunk %u KB lds %zu KB blocks %6u: %.1f GB/s\n", MODE, U, chunk16 * 16 / 1024, lds / 1024, blocks,
           10.0 * n_chunks * chunk16 * 16 / (ms * 1e6));
}

int main(int argc, char **argv)
{
    const unsigned chunk_kb = argc > 1 ? atoi(argv[1]) : 64;
    const size_t lds = (argc > 2 ? atoi(argv[2]) : 24) * 1024;
    const size_t bytes = (size_t) 6 << 30;
    double2 *p; double *out;
    if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) return 1;
    hipMemset(p, 0, bytes);
    const unsigned chunk16 = chunk_kb * 1024 / 16;
    const size_t n_chunks = bytes / 16 / chunk16;
    const unsigned blocks1 = (unsigned) (((n_chunks + 7) / 8) * 8);
    for (int rep = 0; rep < 2; ++rep) {
        run<4, 0>(p, n_chunks, chunk16, out, 4096, 0);
        run<4, 0>(p, n_chunks, chunk16, out, 4096, lds);
        run<2, 1>(p, n_chunks, chunk16, out, blocks1, lds);
        run<4, 1>(p, n_chunks, chunk16, out, blocks1, lds);
        run<8, 1>(p, n_chunks, chunk16, out, blocks1, lds);
        run<4, 2>(p, n_chunks, chunk16, out, (unsigned) n_chunks, lds);
        run<8, 2>(p, n_chunks, chunk16, out, (unsigned) n_chunks, lds);
    }
    return 0;
}
