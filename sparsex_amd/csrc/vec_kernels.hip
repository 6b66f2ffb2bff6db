// vec_kernels.hip -- HBM-resident vectors and their BLAS-1 helpers (gfx950).
//
// Device counterparts of the reference's host vector routines
// (src/internals/Vector.cpp:206-394: VecInit, VecScale, VecScaleAdd, VecAdd,
// VecSub, VecMult, VecCopy) behind the spx_hip_vec_* entry points of
// include/sparsex_hip.h.  All of them are pure streaming kernels (HBM-bound,
// 16 bytes per lane and access).  Every workgroup takes ONE contiguous 64 KB
// chunk of each operand, workgroup b on XCD b % 8 and every XCD walking its own
// contiguous eighth of the vector -- the access pattern of the SpMV kernels, which
// reads 6.3-6.5 TB/s where a grid-stride sweep of a few thousand workgroups reads
// 5.5-6.1 (tools/micro/stream_pattern.hip, profiles/r05/ablation.md).  The dot
// product reduces per wavefront with DPP shuffles, per block through LDS and across
// blocks in a second tiny kernel, in a fixed order (bitwise reproducible).
#include <sparsex_hip.h>

#include "common.hpp"

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <string>

using namespace spx;

struct spx_hip_vec {
    double *data;
    size_t size;
    double *partials;    // dot-product scratch (one double per block)
    int device;
};

namespace {

constexpr int VEC_BLOCK = 256;
constexpr unsigned VEC_CHUNK_MAX = 4096;  // double2 per workgroup and operand: 64 KB (the dot product: reads only)
constexpr unsigned VEC_CHUNK_MAP = 1024;  // ... 16 KB: four accesses per lane (kernels that write; measured best, tools/vec_bench.py)
constexpr unsigned VEC_CHUNK_MIN = 1024;
constexpr unsigned VEC_DOT_BLOCKS = 4096; // most workgroups of the dot product (each ends with a store)
constexpr unsigned VEC_MIN_BLOCKS = 2048; // a vector is cut finely enough for eight workgroups per CU where it is long enough

// chunks of a vector of n doubles, and the grid that covers them: 8 x the chunks of an XCD's eighth
struct VecGrid {
    size_t nchunks, per;
    unsigned blocks, chunk;
};

inline VecGrid grid_for(size_t n, unsigned max_chunk)
{
    VecGrid g;
    g.chunk = max_chunk;
    // (chunk sizes and the dot product's workgroup bound were swept in round 5: profiles/r05/vec_bench_raw.txt)
    while (g.chunk > VEC_CHUNK_MIN && (n / 2 + g.chunk - 1) / g.chunk < VEC_MIN_BLOCKS) g.chunk /= 2;
    g.nchunks = (n / 2 + g.chunk - 1) / g.chunk;
    if (g.nchunks < 1) g.nchunks = 1;
    g.per = (g.nchunks + 7) / 8;
    g.blocks = (unsigned) (g.per * 8);
    return g;
}

// doubles of dot-product scratch for a vector of n doubles: one per workgroup, and the result behind them
inline size_t partials_for(size_t n) { return (n / 2 + 255) / 256 + 16; }     // (the finest cut there is)

// the [lo, hi) range of double2 elements of workgroup b; false: no chunk
__device__ __forceinline__ bool vec_chunk(size_t n2, size_t nchunks, size_t per, unsigned chunk_len, size_t &lo, size_t &hi)
{
    const size_t chunk = (size_t) (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per || chunk >= nchunks) return false;
    lo = chunk * chunk_len;
    hi = lo + chunk_len < n2 ? lo + chunk_len : n2;
    return lo < hi;
}

// kind: 0 init (d = s), 1 scale (d = s*a), 2 axpy (d = a + s*b), 3 copy (d = a)
template <int KIND>
__device__ __forceinline__ double2 vec_map_one(double2 va, double2 vb, double s)
{
    if (KIND == 0) return make_double2(s, s);
    if (KIND == 1) return make_double2(s * va.x, s * va.y);
    if (KIND == 3) return va;
    return make_double2(va.x + s * vb.x, va.y + s * vb.y);
}

template <int KIND>
__global__ __launch_bounds__(VEC_BLOCK) void vec_map_kernel(double *__restrict__ d,
                                                            const double *__restrict__ a,
                                                            const double *__restrict__ b,
                                                            double s, size_t n, size_t nchunks, size_t per, unsigned chunk_len)
{
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const size_t i = n - 1;
        d[i] = KIND == 0 ? s : KIND == 1 ? s * a[i] : KIND == 3 ? a[i] : a[i] + s * b[i];
    }
    size_t lo, hi;
    if (!vec_chunk(n / 2, nchunks, per, chunk_len, lo, hi)) return;
    const double2 *a2 = reinterpret_cast<const double2 *>(a), *b2 = reinterpret_cast<const double2 *>(b);
    double2 *d2 = reinterpret_cast<double2 *>(d);
    size_t i = lo + threadIdx.x;
    // four accesses per operand in flight
    for (; i + 3 * VEC_BLOCK < hi; i += 4 * VEC_BLOCK) {
        double2 va[4], vb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            va[k] = KIND == 0 ? make_double2(0.0, 0.0) : a2[i + k * VEC_BLOCK];
            vb[k] = KIND == 2 ? b2[i + k * VEC_BLOCK] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) d2[i + k * VEC_BLOCK] = vec_map_one<KIND>(va[k], vb[k], s);
    }
    for (; i < hi; i += VEC_BLOCK) {
        const double2 va = KIND == 0 ? make_double2(0.0, 0.0) : a2[i];
        const double2 vb = KIND == 2 ? b2[i] : make_double2(0.0, 0.0);
        d2[i] = vec_map_one<KIND>(va, vb, s);
    }
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
    return v;
}

// SAME: a vector with itself (a norm): every line is loaded once
template <bool SAME>
__global__ __launch_bounds__(VEC_BLOCK) void vec_dot_kernel(const double *__restrict__ a,
                                                            const double *__restrict__ b,
                                                            double *__restrict__ partials, size_t n,
                                                            size_t nchunks, size_t per, unsigned chunk_len)
{
    __shared__ double wsum[VEC_BLOCK / 64];
    double acc = 0.0;
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) acc = a[n - 1] * b[n - 1];
    // the workgroup's chunks: slot, slot + slots, ... of its XCD's eighth (gridDim.x / 8 slots per XCD; every
    // workgroup ends with one store, and stores are dear next to a stream of reads: profiles/r05/ablation.md
    // section 1b -- hence a bounded number of workgroups that each take several chunks)
    const double2 *a2 = reinterpret_cast<const double2 *>(a), *b2 = reinterpret_cast<const double2 *>(b);
    const size_t n2 = n / 2, slots = gridDim.x >> 3;
    for (size_t c = blockIdx.x >> 3; c < per; c += slots) {
        const size_t chunk = (size_t) (blockIdx.x & 7u) * per + c;
        if (chunk >= nchunks) break;
        const size_t lo = chunk * chunk_len, hi = lo + chunk_len < n2 ? lo + chunk_len : n2;
        size_t i = lo + threadIdx.x;
        for (; i + 3 * VEC_BLOCK < hi; i += 4 * VEC_BLOCK) {
            double2 va[4], vb[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                va[k] = a2[i + k * VEC_BLOCK];
                vb[k] = SAME ? va[k] : b2[i + k * VEC_BLOCK];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc = fma(va[k].x, vb[k].x, acc);
                acc = fma(va[k].y, vb[k].y, acc);
            }
        }
        for (; i < hi; i += VEC_BLOCK) {
            const double2 va = a2[i], vb = SAME ? va : b2[i];
            acc = fma(va.x, vb.x, acc);
            acc = fma(va.y, vb.y, acc);
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < VEC_BLOCK / 64; ++w) t += wsum[w];
        partials[blockIdx.x] = t;            // (a workgroup without a chunk: 0)
    }
}

// the workgroups' sums in workgroup order (thread t: t, t + 256, ...; then the wavefronts, then the block)
__global__ __launch_bounds__(VEC_BLOCK) void vec_dot_final_kernel(double *partials, unsigned nblocks)
{
    __shared__ double wsum[VEC_BLOCK / 64];
    double acc = 0.0;
    for (unsigned i = threadIdx.x; i < nblocks; i += VEC_BLOCK) acc += partials[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < VEC_BLOCK / 64; ++w) t += wsum[w];
        partials[nblocks] = t;
    }
}

bool ok(hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    log_msg(LOG_ERR, "HIP failure: %s: %s\n", what, hipGetErrorString(e));
    return false;
}

#define VEC_TRY(expr)                       \
    do {                                    \
        if (!ok((expr), #expr)) {           \
            SETERROR_0(SPX_ERR_VEC);        \
            return SPX_FAILURE;             \
        }                                   \
    } while (0)

spx_error_t same_size(const spx_hip_vec_t *a, const spx_hip_vec_t *b)
{
    if (!a || !b) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid device vector"); return SPX_FAILURE; }
    if (a->size != b->size) { SETERROR_0(SPX_ERR_VEC_DIM); return SPX_FAILURE; }
    return SPX_SUCCESS;
}

template <int KIND>
spx_error_t map(spx_hip_vec_t *d, const spx_hip_vec_t *a, const spx_hip_vec_t *b, double s,
                void *stream)
{
    if (d->size == 0) return SPX_SUCCESS;
    const VecGrid g = grid_for(d->size, VEC_CHUNK_MAP);
    hipLaunchKernelGGL(vec_map_kernel<KIND>, dim3(g.blocks), dim3(VEC_BLOCK), 0,
                       static_cast<hipStream_t>(stream), d->data, a ? a->data : nullptr,
                       b ? b->data : nullptr, s, d->size, g.nchunks, g.per, g.chunk);
    VEC_TRY(hipGetLastError());
    return SPX_SUCCESS;
}

}  // namespace

extern "C" {

spx_hip_vec_t *spx_hip_vec_create(size_t size)
{
    spx_hip_vec_t *v = new spx_hip_vec;
    v->size = size;
    v->data = nullptr;
    v->partials = nullptr;
    if (!ok(hipGetDevice(&v->device), "hipGetDevice") ||
        !ok(hipMalloc(reinterpret_cast<void **>(&v->data), (size ? size : 1) * sizeof(double)),
            "hipMalloc") ||
        !ok(hipMemset(v->data, 0, (size ? size : 1) * sizeof(double)), "hipMemset") ||
        !ok(hipMalloc(reinterpret_cast<void **>(&v->partials),
                      partials_for(size) * sizeof(double)), "hipMalloc")) {
        if (v->data) (void) hipFree(v->data);
        delete v;
        SETERROR_1(SPX_ERR_VEC, "device vector allocation failed (no usable HIP device?)");
        return NULL;
    }
    return v;
}

spx_hip_vec_t *spx_hip_vec_create_from_host(const spx_vector_t *h)
{
    if (!h) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid vector"); return NULL; }
    spx_hip_vec_t *v = spx_hip_vec_create(h->size);
    if (v && spx_hip_vec_upload(v, h, NULL) != SPX_SUCCESS) {
        spx_hip_vec_destroy(v);
        return NULL;
    }
    return v;
}

spx_error_t spx_hip_vec_destroy(spx_hip_vec_t *v)
{
    if (!v) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid device vector"); return SPX_FAILURE; }
    (void) hipFree(v->data);
    (void) hipFree(v->partials);
    delete v;
    return SPX_SUCCESS;
}

spx_value_t *spx_hip_vec_data(spx_hip_vec_t *v) { return v ? v->data : NULL; }
size_t spx_hip_vec_size(const spx_hip_vec_t *v) { return v ? v->size : 0; }

spx_error_t spx_hip_vec_upload(spx_hip_vec_t *dst, const spx_vector_t *src, void *stream)
{
    if (!dst || !src) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid vector"); return SPX_FAILURE; }
    if (dst->size != src->size) { SETERROR_0(SPX_ERR_VEC_DIM); return SPX_FAILURE; }
    VEC_TRY(hipMemcpyAsync(dst->data, src->elements, src->size * sizeof(double),
                           hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
    VEC_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return SPX_SUCCESS;
}

spx_error_t spx_hip_vec_download(const spx_hip_vec_t *src, spx_vector_t *dst, void *stream)
{
    if (!dst || !src) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid vector"); return SPX_FAILURE; }
    if (dst->size != src->size) { SETERROR_0(SPX_ERR_VEC_DIM); return SPX_FAILURE; }
    VEC_TRY(hipMemcpyAsync(dst->elements, src->data, src->size * sizeof(double),
                           hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    VEC_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return SPX_SUCCESS;
}

spx_error_t spx_hip_vec_init(spx_hip_vec_t *v, spx_value_t val, void *stream)
{
    if (!v) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid device vector"); return SPX_FAILURE; }
    return map<0>(v, nullptr, nullptr, val, stream);
}

spx_error_t spx_hip_vec_scale(const spx_hip_vec_t *v1, spx_hip_vec_t *v2, spx_value_t num,
                              void *stream)
{
    if (same_size(v1, v2) != SPX_SUCCESS) return SPX_FAILURE;
    return map<1>(v2, v1, nullptr, num, stream);
}

spx_error_t spx_hip_vec_scale_add(const spx_hip_vec_t *v1, const spx_hip_vec_t *v2,
                                  spx_hip_vec_t *v3, spx_value_t num, void *stream)
{
    if (same_size(v1, v2) != SPX_SUCCESS || same_size(v1, v3) != SPX_SUCCESS) return SPX_FAILURE;
    return map<2>(v3, v1, v2, num, stream);
}

spx_error_t spx_hip_vec_add(const spx_hip_vec_t *v1, const spx_hip_vec_t *v2, spx_hip_vec_t *v3,
                            void *stream)
{
    return spx_hip_vec_scale_add(v1, v2, v3, 1.0, stream);
}

spx_error_t spx_hip_vec_sub(const spx_hip_vec_t *v1, const spx_hip_vec_t *v2, spx_hip_vec_t *v3,
                            void *stream)
{
    return spx_hip_vec_scale_add(v1, v2, v3, -1.0, stream);
}

spx_error_t spx_hip_vec_copy(const spx_hip_vec_t *v1, spx_hip_vec_t *v2, void *stream)
{
    if (same_size(v1, v2) != SPX_SUCCESS) return SPX_FAILURE;
    return map<3>(v2, v1, nullptr, 0.0, stream);
}

spx_error_t spx_hip_vec_mul(const spx_hip_vec_t *v1, const spx_hip_vec_t *v2, spx_value_t *result,
                            void *stream_)
{
    if (same_size(v1, v2) != SPX_SUCCESS) return SPX_FAILURE;
    if (!result) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid result pointer"); return SPX_FAILURE; }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    VecGrid g = grid_for(v1->size, VEC_CHUNK_MAX);
    if (g.blocks > (unsigned) VEC_DOT_BLOCKS) g.blocks = (unsigned) VEC_DOT_BLOCKS & ~7u;
    if (v1->data == v2->data)
        hipLaunchKernelGGL(vec_dot_kernel<true>, dim3(g.blocks), dim3(VEC_BLOCK), 0, stream, v1->data,
                           v2->data, v1->partials, v1->size, g.nchunks, g.per, g.chunk);
    else
        hipLaunchKernelGGL(vec_dot_kernel<false>, dim3(g.blocks), dim3(VEC_BLOCK), 0, stream, v1->data,
                           v2->data, v1->partials, v1->size, g.nchunks, g.per, g.chunk);
    hipLaunchKernelGGL(vec_dot_final_kernel, dim3(1), dim3(VEC_BLOCK), 0, stream, v1->partials, g.blocks);
    VEC_TRY(hipGetLastError());
    VEC_TRY(hipMemcpyAsync(result, v1->partials + g.blocks, sizeof(double),
                           hipMemcpyDeviceToHost, stream));
    VEC_TRY(hipStreamSynchronize(stream));
    return SPX_SUCCESS;
}

// ---- the roof of a read stream with a few stores in it (diagnostic) -------------------------------------
// What the SpMV kernels ask of the memory system is not a pure read: every workgroup ends with the stores of its
// rows of y.  This probe has their shape -- workgroup b on XCD b % 8, a contiguous chunk of `src` read per
// workgroup with 16-byte loads, then `wr` doubles stored to the workgroup's stretch of `dst` -- so that a
// product's achieved rate can be set against a roof with the SAME read / write mix, measured on the same box
// (bench.py: roofline.measured_mixed_peak; tools/micro/stream_pattern.hip mode 6 is the stand-alone form).
__global__ __launch_bounds__(256) void vec_probe_rw_kernel(const double2 *src, size_t n_chunks, size_t per, unsigned chunk16,
                                                           double *dst, unsigned wr)
{
    const size_t slot = blockIdx.x >> 3;
    const size_t chunk = (size_t) (blockIdx.x & 7u) * per + slot;
    if (slot >= per || chunk >= n_chunks) return;
    const double2 *q = src + chunk * chunk16;
    double acc = 0.0;
    for (unsigned i = threadIdx.x; i + 3u * 256u < chunk16; i += 4u * 256u) {
        const double2 a = q[i], b = q[i + 256u], c = q[i + 512u], d = q[i + 768u];
        acc += (a.x + a.y) + (b.x + b.y) + (c.x + c.y) + (d.x + d.y);
    }
    __syncthreads();
    double *out = dst + chunk * wr;
    for (unsigned i = threadIdx.x; i < wr; i += 256u) out[i] = acc;
}

spx_error_t spx_hip_probe_read_write(const spx_hip_vec_t *src, spx_hip_vec_t *dst, size_t chunk_doubles,
                                     size_t write_doubles, void *stream_)
{
    if (!src || !dst) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid device vector"); return SPX_FAILURE; }
    if (chunk_doubles < 2048 || chunk_doubles % 2048 || write_doubles > chunk_doubles) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "chunk: a multiple of 2048 doubles; no more doubles written than read");
        return SPX_FAILURE;
    }
    const size_t n_chunks = src->size / chunk_doubles;
    if (n_chunks == 0 || dst->size < n_chunks * write_doubles) {
        SETERROR_1(SPX_ERR_ARG_INVALID, "vectors too short for one chunk / for the stores of every chunk");
        return SPX_FAILURE;
    }
    const size_t per = (n_chunks + 7) / 8;
    hipLaunchKernelGGL(vec_probe_rw_kernel, dim3((unsigned) (per * 8)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                       reinterpret_cast<const double2 *>(src->data), n_chunks, per, (unsigned) (chunk_doubles / 2), dst->data,
                       (unsigned) write_doubles);
    VEC_TRY(hipGetLastError());
    return SPX_SUCCESS;
}

spx_error_t spx_hip_matvec_kernel_vec(spx_value_t alpha, const spx_matrix_t *A,
                                      const spx_hip_vec_t *x, spx_value_t beta,
                                      spx_hip_vec_t *y, void *stream)
{
    if (!x || !y) { SETERROR_1(SPX_ERR_ARG_INVALID, "invalid device vector"); return SPX_FAILURE; }
    if (A && (x->size != (size_t) spx_mat_get_ncols(A) || y->size != (size_t) spx_mat_get_nrows(A))) {
        SETERROR_0(SPX_ERR_DIM);
        return SPX_FAILURE;
    }
    return spx_hip_matvec_kernel(alpha, A, x->data, beta, y->data, stream);
}

}  // extern "C"
