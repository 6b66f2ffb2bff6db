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
//   mode 6  as mode 1, and every workgroup ends with 304 doubles written (a row-block's rows of y: 2.4 KB per
//           64 KB read): plain stores, non-temporal stores, stores into a small region that stays in the L2,
//           and the same bytes written by one workgroup in 64 (64 x 304 doubles at once)
//           (round 6: also `sc1` / `sc0 sc1` stores, 8 and 16 bytes per lane)
//   mode 7  as mode 6 with plain stores, but a workgroup takes K consecutive chunks of its XCD's list one after the
//           other and writes each one's 304 doubles as it goes (1 / K as many workgroups, the same stores)
//   mode 5  as mode 1, and every workgroup also reads XKB (argv[3], default 24) kilobytes of a region small enough
//           to stay in its XCD's L2 (what the staging of x asks of the L2-to-L1 path on top of the stream)
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


// mode 6: the stream as in mode 1, then the workgroup's rows of y
//   STORE 0 none, 1 plain, 2 non-temporal, 3 plain into an L2-resident region, 4 every 64th workgroup writes 64 tiles,
//   5 tiles of 298 doubles packed one behind the other (they begin and end inside 128-byte lines)
template <int U, int STORE>
__global__ __launch_bounds__(256) void pattern_wr_kernel(const double2 *p, size_t n_chunks, unsigned chunk16, double *y, double *out)
{
    extern __shared__ double lds[];
    double acc = 0.0;
    const size_t per = (n_chunks + 7) / 8;
    const size_t chunk = (size_t) (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per || chunk >= n_chunks) return;
    const double2 *q = p + chunk * chunk16;
    for (unsigned i = threadIdx.x; i + (U - 1) * 256u < chunk16; i += U * 256u) {
        double2 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = q[i + k * 256u];
#pragma unroll
        for (int k = 0; k < U; ++k) acc += v[k].x + v[k].y;
    }
    __syncthreads();
    if (STORE == 1 || STORE == 2 || STORE == 3) {
        double *dst = y + (STORE == 3 ? (size_t) (blockIdx.x & 255u) * 304u : chunk * 304u);
        for (unsigned i = threadIdx.x; i < 304u; i += 256u) {
            if (STORE == 2) __builtin_nontemporal_store(acc, dst + i);
            else dst[i] = acc;
        }
    } else if (STORE == 8 || STORE == 9) {
        // the store flavours of MI355X_MICROARCH.md that DROP the line from the XCD's L2 instead of keeping it:
        // `sc1` (8) and `sc0 sc1` (9), 8 bytes per lane as the kernels' write-out issues them
        double *dst = y + chunk * 304u;
        for (unsigned i = threadIdx.x; i < 304u; i += 256u) {
            if (STORE == 8) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(dst + i), "v"(acc) : "memory");
            else asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(dst + i), "v"(acc) : "memory");
        }
    } else if (STORE == 10 || STORE == 11) {
        // ... and the same as 16 bytes per lane (152 lanes): plain (10) and sc1 (11)
        double *dst = y + chunk * 304u;
        const double2 v2 = double2{acc, acc};
        typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
        const unsigned lo = (unsigned) __double2loint(acc), hi = (unsigned) __double2hiint(acc);
        const v4u_t w4 = {lo, hi, lo, hi};
        for (unsigned i = threadIdx.x; i < 152u; i += 256u) {
            if (STORE == 10) reinterpret_cast<double2 *>(dst)[i] = v2;
            else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(reinterpret_cast<double2 *>(dst) + i), "v"(w4) : "memory");
        }
    } else if (STORE == 12) {
        // the same 304 doubles added atomically (global_atomic_add_f64, no return): the symmetric kernels' hand-over
        double *dst = y + chunk * 304u;
        for (unsigned i = threadIdx.x; i < 304u; i += 256u) unsafeAtomicAdd(dst + i, acc);
    } else if (STORE == 6) {
        if (threadIdx.x == 0) y[chunk] = acc;                    // one lane, 8 bytes (a dot product's partial sum)
    } else if (STORE == 7) {
        if (threadIdx.x == 0) atomicAdd(y + (chunk & 1023u), acc);   // ... as an atomic add to one of 1024 doubles
    } else if (STORE == 5) {
        double *dst = y + chunk * 298u + 1u;
        for (unsigned i = threadIdx.x; i < 298u; i += 256u) dst[i] = acc;
    } else if (STORE == 4) {
        if (((blockIdx.x >> 3) & 63u) == 0u) {
            double *dst = y + chunk * 304u;
            for (unsigned i = threadIdx.x; i < 64u * 304u; i += 256u) dst[i] = acc;
        }
    }
    if (acc == 1.2345) out[0] = acc + lds[0];
}

template <int STORE>
static void run_wr(const double2 *p, size_t n_chunks, unsigned chunk16, double *y, double *out, unsigned blocks, size_t lds)
{
    hipEvent_t e0, e1;
    (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    (void) hipFuncSetAttribute(reinterpret_cast<const void *>(&pattern_wr_kernel<4, STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    pattern_wr_kernel<4, STORE><<<blocks, 256, lds>>>(p, n_chunks, chunk16, y, out);
    (void) hipDeviceSynchronize();
    (void) hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) pattern_wr_kernel<4, STORE><<<blocks, 256, lds>>>(p, n_chunks, chunk16, y, out);
    (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
    float ms; (void) hipEventElapsedTime(&ms, e0, e1);
    static const char *what[] = {"no stores", "plain stores", "non-temporal stores", "stores into an L2-resident region",
                                 "one workgroup in 64 stores 64 tiles", "298 doubles, tiles packed (partial lines)",
                                 "ONE lane stores 8 bytes", "one lane adds atomically to 1 of 1024",
                                 "sc1 stores (8 bytes per lane)", "sc0 sc1 stores (8 bytes per lane)",
                                 "plain stores, 16 bytes per lane", "sc1 stores, 16 bytes per lane",
                                 "atomic adds (8 bytes per lane)"};
    printf("mode 6 U=4 chunk %u KB, 2.4 KB written per chunk, %-36s: stream %.1f GB/s\n", chunk16 * 16 / 1024, what[STORE],
           10.0 * n_chunks * chunk16 * 16 / (ms * 1e6));
}

// mode 7: K chunks per workgroup, each followed by its stores
template <int U>
__global__ __launch_bounds__(256) void pattern_multi_kernel(const double2 *p, size_t n_chunks, unsigned chunk16, double *y,
                                                            unsigned K, double *out)
{
    extern __shared__ double lds[];
    const size_t per = (n_chunks + 7) / 8;
    double acc = 0.0;
    for (unsigned k = 0; k < K; ++k) {
        const size_t slot = (size_t) (blockIdx.x >> 3) * K + k;
        const size_t chunk = (size_t) (blockIdx.x & 7u) * per + slot;
        if (slot >= per || chunk >= n_chunks) break;
        const double2 *q = p + chunk * chunk16;
        for (unsigned i = threadIdx.x; i + (U - 1) * 256u < chunk16; i += U * 256u) {
            double2 v[U];
#pragma unroll
            for (int j = 0; j < U; ++j) v[j] = q[i + j * 256u];
#pragma unroll
            for (int j = 0; j < U; ++j) acc += v[j].x + v[j].y;
        }
        __syncthreads();
        double *dst = y + chunk * 304u;
        for (unsigned i = threadIdx.x; i < 304u; i += 256u) dst[i] = acc;
    }
    if (acc == 1.2345) out[0] = acc + lds[0];
}

static void run_multi(const double2 *p, size_t n_chunks, unsigned chunk16, double *y, unsigned K, double *out, size_t lds)
{
    hipEvent_t e0, e1;
    (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    (void) hipFuncSetAttribute(reinterpret_cast<const void *>(&pattern_multi_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const size_t per = (n_chunks + 7) / 8;
    const unsigned blocks = (unsigned) (((per + K - 1) / K) * 8);
    pattern_multi_kernel<4><<<blocks, 256, lds>>>(p, n_chunks, chunk16, y, K, out);
    (void) hipDeviceSynchronize();
    (void) hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) pattern_multi_kernel<4><<<blocks, 256, lds>>>(p, n_chunks, chunk16, y, K, out);
    (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
    float ms; (void) hipEventElapsedTime(&ms, e0, e1);
    printf("mode 7 U=4 chunk %u KB, 2.4 KB written per chunk, %2u chunks per workgroup (%6u workgroups): stream %.1f GB/s\n",
           chunk16 * 16 / 1024, K, blocks, 10.0 * n_chunks * chunk16 * 16 / (ms * 1e6));
}

// mode 5: the stream as in mode 1 plus xkb kilobytes per workgroup from an L2-resident region
template <int U>
__global__ __launch_bounds__(256) void pattern_l2_kernel(const double2 *p, size_t n_chunks, unsigned chunk16, const double2 *hot,
                                                         unsigned xkb, double *out)
{
    extern __shared__ double lds[];
    double acc = 0.0;
    const size_t per = (n_chunks + 7) / 8;
    const size_t chunk = (size_t) (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per || chunk >= n_chunks) return;
    // the workgroup's piece of the hot region: 32 pieces per XCD, reused by every 32nd workgroup of that XCD
    const double2 *h = hot + ((size_t) (blockIdx.x & 7u) * 32u + ((blockIdx.x >> 3) & 31u)) * (size_t) (xkb * 64u);
    for (unsigned i = threadIdx.x; i < xkb * 64u; i += 256u) {
        const double2 v = h[i];
        acc += v.x + v.y;
    }
    const double2 *q = p + chunk * chunk16;
    for (unsigned i = threadIdx.x; i + (U - 1) * 256u < chunk16; i += U * 256u) {
        double2 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = q[i + k * 256u];
#pragma unroll
        for (int k = 0; k < U; ++k) acc += v[k].x + v[k].y;
    }
    if (acc == 1.2345) out[0] = acc + lds[0];
}

static void run_l2(const double2 *p, size_t n_chunks, unsigned chunk16, const double2 *hot, unsigned xkb, double *out,
                   unsigned blocks, size_t lds)
{
    hipEvent_t e0, e1;
    (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    (void) hipFuncSetAttribute(reinterpret_cast<const void *>(&pattern_l2_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    pattern_l2_kernel<4><<<blocks, 256, lds>>>(p, n_chunks, chunk16, hot, xkb, out);
    (void) hipDeviceSynchronize();
    (void) hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) pattern_l2_kernel<4><<<blocks, 256, lds>>>(p, n_chunks, chunk16, hot, xkb, out);
    (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
    float ms; (void) hipEventElapsedTime(&ms, e0, e1);
    const double stream = 10.0 * n_chunks * chunk16 * 16, l2 = 10.0 * n_chunks * xkb * 1024.0;
    printf("mode 5 U=4 chunk %u KB + %2u KB from the L2, lds %zu KB: stream %.1f GB/s, stream + L2 reads %.1f GB/s\n",
           chunk16 * 16 / 1024, xkb, lds / 1024, stream / (ms * 1e6), (stream + l2) / (ms * 1e6));
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
    // mode 6: y is a buffer of its own
    {
        double *y;
        if (hipMalloc(&y, (n_chunks + 64) * 304 * sizeof(double)) != hipSuccess) return 1;
        for (int rep = 0; rep < 2; ++rep) {
            run_wr<0>(p, n_chunks, chunk16, y, out, blocks1, lds);
            run_wr<1>(p, n_chunks, chunk16, y, out, blocks1, lds);
            run_wr<2>(p, n_chunks, chunk16, y, out, blocks1, lds);
            run_wr<3>(p, n_chunks, chunk16, y, out, blocks1, lds);
            run_wr<4>(p, n_chunks, chunk16, y, out, blocks1, lds);
            run_wr<5>(p, n_chunks, chunk16, y, out, blocks1, lds);
            run_wr<6>(p, n_chunks, chunk16, y, out, blocks1, lds);
            run_wr<7>(p, n_chunks, chunk16, y, out, blocks1, lds);
            run_wr<8>(p, n_chunks, chunk16, y, out, blocks1, lds);
            run_wr<9>(p, n_chunks, chunk16, y, out, blocks1, lds);
            run_wr<10>(p, n_chunks, chunk16, y, out, blocks1, lds);
            run_wr<11>(p, n_chunks, chunk16, y, out, blocks1, lds);
            run_wr<12>(p, n_chunks, chunk16, y, out, blocks1, lds);
            run_wr<1>(p, n_chunks, chunk16, y, out, blocks1, lds);
        }
        for (unsigned K : {1u, 2u, 4u, 8u, 24u}) run_multi(p, n_chunks, chunk16, y, K, out, lds);
        (void) hipFree(y);
        // ... and y in memory allocated otherwise: fine-grained, uncached
        for (unsigned flag : {(unsigned) hipDeviceMallocFinegrained, (unsigned) hipDeviceMallocUncached}) {
            double *y2 = nullptr;
            if (hipExtMallocWithFlags(reinterpret_cast<void **>(&y2), (n_chunks + 64) * 304 * sizeof(double), flag) != hipSuccess) {
                printf("hipExtMallocWithFlags(%u) failed\n", flag);
                (void) hipGetLastError();
                continue;
            }
            printf("y allocated with hipExtMallocWithFlags(%s):\n", flag == (unsigned) hipDeviceMallocFinegrained ? "hipDeviceMallocFinegrained" : "hipDeviceMallocUncached");
            run_wr<0>(p, n_chunks, chunk16, y2, out, blocks1, lds);
            run_wr<1>(p, n_chunks, chunk16, y2, out, blocks1, lds);
            run_wr<2>(p, n_chunks, chunk16, y2, out, blocks1, lds);
            (void) hipFree(y2);
        }
    }
    // mode 5: the hot region is the last 64 MB of the buffer; the stream covers the rest
    {
        const size_t hot_bytes = (size_t) 64 << 20;
        const size_t n_str = (bytes - hot_bytes) / 16 / chunk16;
        const double2 *hot = p + (bytes - hot_bytes) / 16;
        const unsigned b5 = (unsigned) (((n_str + 7) / 8) * 8);
        for (unsigned xkb : {0u, 8u, 16u, 24u, 32u, 48u})
            if ((size_t) 8 * 32 * xkb * 1024 <= hot_bytes) run_l2(p, n_str, chunk16, hot, xkb, out, b5, lds);
    }
    return 0;
}
