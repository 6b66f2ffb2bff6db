// Microbenchmark: what a read-only stream reaches when it is read the way the SpMV kernels read the
// values of a matrix -- every workgroup its own contiguous chunk (a row-block's 64 KB), workgroup b on
// XCD b % 8, each XCD walking its own contiguous eighth of the buffer -- against the grid-stride sweep
// of tools/micro/stream_read.hip in which the whole chip moves through the buffer together.
//   mode 0  grid-stride sweep (the "read roof" probe)
//   mode 1  chunk per workgroup, XCD x owns the x-th eighth of the chunks (the SpMV launch order)
//   mode 2  chunk per workgroup, chunks dealt round robin over the XCDs (chunk = blockIdx)
//   mode 3  as mode 1, but the chunk is read as the values of width-3 passes are: pieces of 1536 bytes per
//           wavefront, a 16-byte load per lane on the first kilobyte and an 8-byte load per lane on the rest
//   mode 4  as mode 3 behind one dependent 8-byte load per workgroup (the row-block header's round trip)
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
        if (MODE >= 3) {
            if (MODE == 4) {
                // (the buffer holds zeros: the loaded offset is 0, but the loads below wait for it)
                const unsigned long long off = reinterpret_cast<const unsigned long long *>(p)[chunk * 8];
                q += off;
            }
            const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
            const unsigned n_pieces = chunk16 * 16u / 1536u;
            const char *base = reinterpret_cast<const char *>(q);
            for (unsigned i = wave; i + (U - 1) * 4u < n_pieces; i += U * 4u) {
                double2 a[U];
                double b[U];
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    const char *piece = base + (size_t) (i + k * 4u) * 1536u;
                    a[k] = *reinterpret_cast<const double2 *>(piece + lane * 16u);
                    b[k] = *reinterpret_cast<const double *>(piece + 1024u + lane * 8u);
                }
#pragma unroll
                for (int k = 0; k < U; ++k) acc += a[k].x + a[k].y + b[k];
            }
            if (acc == 1.2345) out[0] = acc + lds[0];
            return;
        }
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
    // (modes 3, 4 read whole pieces of 1536 bytes, U x 4 at a time: count what was read)
    double per_chunk = (double) chunk16 * 16;
    if (MODE >= 3) {
        const unsigned n_pieces = chunk16 * 16u / 1536u;
        unsigned got = 0;
        for (unsigned w = 0; w < 4; ++w)
            for (unsigned i = w; i + (U - 1) * 4u < n_pieces; i += U * 4u) got += U;
        per_chunk = 1536.0 * got;
    }
    printf("mode %d U=%d chunk %u KB lds %zu KB blocks %6u: %.1f GB/s\n", MODE, U, chunk16 * 16 / 1024, lds / 1024, blocks,
           10.0 * n_chunks * per_chunk / (ms * 1e6));
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
        run<2, 3>(p, n_chunks, chunk16, out, blocks1, lds);
        run<4, 3>(p, n_chunks, chunk16, out, blocks1, lds);
        run<2, 4>(p, n_chunks, chunk16, out, blocks1, lds);
        run<4, 4>(p, n_chunks, chunk16, out, blocks1, lds);
    }
    return 0;
}
