// vec_kernels.hip -- HBM-resident vectors and their BLAS-1 helpers (gfx950).
//
// Device counterparts of the reference's host vector routines
// (src/internals/Vector.cpp:206-394: VecInit, VecScale, VecScaleAdd, VecAdd,
// VecSub, VecMult, VecCopy) behind the spx_hip_vec_* entry points of
// include/sparsex_hip.h.  All of them are pure streaming kernels (HBM-bound,
// 16 bytes per lane and access, grid-stride); the dot product reduces per
// wavefront with DPP shuffles, per block through LDS and across blocks in a
// second tiny kernel, in a fixed order (bitwise reproducible).
#include <sparsex_hip.h>

#include "common.hpp"

#include <hip/hip_runtime.h>

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
constexpr int VEC_MAX_BLOCKS = 2048;    // 256 CUs x 8 blocks

inline unsigned grid_for(size_t n)
{
    size_t per_block = (size_t) VEC_BLOCK * 2;          // two doubles per thread and step
    size_t b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > (size_t) VEC_MAX_BLOCKS) b = VEC_MAX_BLOCKS;
    return (unsigned) b;
}

// kind: 0 init (d = s), 1 scale (d = s*a), 2 axpy (d = a + s*b), 3 copy (d = a)
template <int KIND>
__global__ __launch_bounds__(VEC_BLOCK) void vec_map_kernel(double *__restrict__ d,
                                                            const double *__restrict__ a,
                                                            const double *__restrict__ b,
                                                            double s, size_t n)
{
    const size_t n2 = n / 2;
    const size_t stride = (size_t) gridDim.x * VEC_BLOCK;
    for (size_t i = (size_t) blockIdx.x * VEC_BLOCK + threadIdx.x; i < n2; i += stride) {
        double2 r;
        if (KIND == 0) {
            r = make_double2(s, s);
        } else {
            const double2 va = reinterpret_cast<const double2 *>(a)[i];
            if (KIND == 1) r = make_double2(s * va.x, s * va.y);
            else if (KIND == 3) r = va;
            else {
                const double2 vb = reinterpret_cast<const double2 *>(b)[i];
                r = make_double2(va.x + s * vb.x, va.y + s * vb.y);
            }
        }
        reinterpret_cast<double2 *>(d)[i] = r;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const size_t i = n - 1;
        d[i] = KIND == 0 ? s : KIND == 1 ? s * a[i] : KIND == 3 ? a[i] : a[i] + s * b[i];
    }
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
    return v;
}

__global__ __launch_bounds__(VEC_BLOCK) void vec_dot_kernel(const double *__restrict__ a,
                                                            const double *__restrict__ b,
                                                            double *__restrict__ partials, size_t n)
{
    __shared__ double wsum[VEC_BLOCK / 64];
    const size_t n2 = n / 2;
    const size_t stride = (size_t) gridDim.x * VEC_BLOCK;
    double acc = 0.0;
    for (size_t i = (size_t) blockIdx.x * VEC_BLOCK + threadIdx.x; i < n2; i += stride) {
        const double2 va = reinterpret_cast<const double2 *>(a)[i];
        const double2 vb = reinterpret_cast<const double2 *>(b)[i];
        acc = fma(va.x, vb.x, acc);
        acc = fma(va.y, vb.y, acc);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) acc = fma(a[n - 1], b[n - 1], acc);
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < VEC_BLOCK / 64; ++w) t += wsum[w];
        partials[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(64) void vec_dot_final_kernel(double *partials, unsigned nblocks)
{
    double acc = 0.0;
    for (unsigned i = threadIdx.x; i < nblocks; i += 64) acc += partials[i];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) partials[VEC_MAX_BLOCKS] = acc;
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
    hipLaunchKernelGGL(vec_map_kernel<KIND>, dim3(grid_for(d->size)), dim3(VEC_BLOCK), 0,
                       static_cast<hipStream_t>(stream), d->data, a ? a->data : nullptr,
                       b ? b->data : nullptr, s, d->size);
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
                      (VEC_MAX_BLOCKS + 1) * sizeof(double)), "hipMalloc")) {
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
    const unsigned blocks = grid_for(v1->size);
    hipLaunchKernelGGL(vec_dot_kernel, dim3(blocks), dim3(VEC_BLOCK), 0, stream, v1->data,
                       v2->data, v1->partials, v1->size);
    hipLaunchKernelGGL(vec_dot_final_kernel, dim3(1), dim3(64), 0, stream, v1->partials, blocks);
    VEC_TRY(hipGetLastError());
    VEC_TRY(hipMemcpyAsync(result, v1->partials + VEC_MAX_BLOCKS, sizeof(double),
                           hipMemcpyDeviceToHost, stream));
    VEC_TRY(hipStreamSynchronize(stream));
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
