// Experiment behind row a20 of the round-1 verdict: does an MFMA formulation of the dense
// 8x8 tiles beat the VALU one?  A stream of N tiles (64 doubles each, the size of syn-nd24k's
// symmetric tile stream) is multiplied both ways:
//   valu_row   r = T * xc                      (what a BLOCK unit of the general path needs)
//   mfma_row   the same through v_mfma_f64_16x16x4 (two tiles side by side per instruction pair,
//              one useful output column each)
//   valu_sym   r = T * xc  and  c = T^T * xr   (the symmetric tile pass: lanes = rows, the column
//              sums by a 4+2+1 exchange among the tile's eight lanes)
//   mfma_sym   both through MFMA: the second product needs the tile in the transposed lane
//              layout, i.e. a second read of its values (L2 hit)
// Every variant reads the same bytes from memory once (mfma_sym: twice, the second time from
// cache) and adds its results into small output vectors; results are checked against the host.
// build: hipcc -w --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/micro/tile_mfma.hip -o /tmp/tile_mfma
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int NX = 72000;          // rows/columns the tiles point into (syn-nd24k)

// ---- VALU: lanes 8t..8t+7 = rows of tile t; values interleaved in column pairs, as in the product
template <bool SYM>
__global__ __launch_bounds__(256) void valu_kernel(const double *vals, const uint32_t *tcol, const uint32_t *trow,
                                                   const double *x, double *y, size_t npass)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t) blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((size_t) gridDim.x * 256) >> 6;
    for (size_t p = wave; p < npass; p += nw) {
        const double *v = vals + p * 512;
        double2 v2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v2[q] = *reinterpret_cast<const double2 *>(v + q * 128 + lane * 2);
        const uint32_t c0 = tcol[p * 8 + (lane >> 3)], r0 = trow[p * 8 + (lane >> 3)];
        const int i = lane & 7;
        const double2 *xp = reinterpret_cast<const double2 *>(x + c0);
        double vv[8], xc[8], t = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double2 xx = xp[q];
            vv[2 * q] = v2[q].x; vv[2 * q + 1] = v2[q].y;
            xc[2 * q] = xx.x; xc[2 * q + 1] = xx.y;
        }
#pragma unroll
        for (int w = 0; w < 8; ++w) t = fma(vv[w], xc[w], t);
        atomicAdd(&y[r0 + i], t);
        if (SYM) {
            const double xr = x[r0 + i];
            double p8[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) p8[w] = vv[w] * xr;
            double p4[4], p2[2], cs;
            { const bool hi = i & 4;
#pragma unroll
              for (int w = 0; w < 4; ++w) p4[w] = (hi ? p8[w + 4] : p8[w]) + __shfl_xor(hi ? p8[w] : p8[w + 4], 4); }
            { const bool hi = i & 2;
#pragma unroll
              for (int w = 0; w < 2; ++w) p2[w] = (hi ? p4[w + 2] : p4[w]) + __shfl_xor(hi ? p4[w] : p4[w + 2], 2); }
            { const bool hi = i & 1; cs = (hi ? p2[1] : p2[0]) + __shfl_xor(hi ? p2[0] : p2[1], 1); }
            atomicAdd(&y[NX + c0 + i], cs);
        }
    }
}

// ---- MFMA: a pass = 8 tiles = 4 pairs; per pair the 128 values are stored in operand order:
// mvals[pair][step s][lane l] = T_{l%16 < 8 ? a : b}[(l%16)%8][4s + l/16]  (A operand: m = l%16, k = l/16).
// SYM additionally reads tvals[pair][s][l] = T_{..}[4s + l/16][(l%16)%8] (the transposed tile) for c = T^T xr.
template <bool SYM, int MAP>
__global__ __launch_bounds__(256) void mfma_kernel(const double *mvals, const double *tvals, const uint32_t *tcol,
                                                   const uint32_t *trow, const double *x, double *y, size_t npass)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t) blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((size_t) gridDim.x * 256) >> 6;
    const int j = lane & 15, k = lane >> 4;
    for (size_t p = wave; p < npass; p += nw) {
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const size_t tile_a = p * 8 + 2 * pr;
            const double *v = mvals + (p * 4 + pr) * 128;
            const double a0 = v[lane], a1 = v[64 + lane];
            // B operand: column 0 = xc of tile a, column 1 = xc of tile b, the rest 0
            const uint32_t c0 = tcol[tile_a + (j & 1)], r0 = trow[tile_a + (j & 1)];
            const double b0 = j < 2 ? x[c0 + k] : 0.0, b1 = j < 2 ? x[c0 + 4 + k] : 0.0;
            double4_t d = {0.0, 0.0, 0.0, 0.0};
            d = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, d, 0, 0, 0);
            // D[m][n = lane%16] with m = 4*(lane/16) + r (MAP 0) or (lane/16) + 4*r (MAP 1; what gfx950
            // does for f64, established by this very check): column 0 rows 0-7 = tile a, column 1 rows 8-15 = tile b
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = MAP ? k + 4 * r : 4 * k + r;
                if (j < 2 && (m >> 3) == j) atomicAdd(&y[r0 + (m & 7)], d[r]);
            }
            if (SYM) {
                const double *tv = tvals + (p * 4 + pr) * 128;
                const double t0 = tv[lane], t1 = tv[64 + lane];
                const double e0 = j < 2 ? x[r0 + k] : 0.0, e1 = j < 2 ? x[r0 + 4 + k] : 0.0;
                double4_t c = {0.0, 0.0, 0.0, 0.0};
                c = __builtin_amdgcn_mfma_f64_16x16x4f64(t0, e0, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f64_16x16x4f64(t1, e1, c, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = MAP ? k + 4 * r : 4 * k + r;
                    if (j < 2 && (m >> 3) == j) atomicAdd(&y[NX + c0 + (m & 7)], c[r]);
                }
            }
        }
    }
}

int main(int argc, char **argv)
{
    const size_t ntiles = argc > 1 ? strtoull(argv[1], 0, 10) : 1800000;
    const size_t npass = ntiles / 8;
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    std::vector<double> T(npass * 512), hv(npass * 512), hm(npass * 512), ht(npass * 512), hx(NX);
    std::vector<uint32_t> hc(npass * 8), hr(npass * 8);
    for (auto &v : T) v = (double) (rnd() % 2001) / 1000.0 - 1.0;
    for (auto &v : hx) v = (double) (rnd() % 2001) / 10000.0 - 0.1;
    for (size_t t = 0; t < npass * 8; ++t) {
        hr[t] = (uint32_t) ((rnd() % (NX / 8 - 1) + 1) * 8);
        hc[t] = (uint32_t) ((rnd() % (hr[t] / 8)) * 8);
    }
    for (size_t p = 0; p < npass; ++p)
        for (int t = 0; t < 8; ++t)
            for (int i = 0; i < 8; ++i)
                for (int w = 0; w < 8; ++w) {
                    const double a = T[(p * 8 + t) * 64 + i * 8 + w];
                    hv[p * 512 + (w >> 1) * 128 + (t * 8 + i) * 2 + (w & 1)] = a;      // product layout
                    const int pr = t >> 1, half = t & 1;
                    // A operand of the row product: lane l = 16*(w%4) + 8*half + i, step w/4
                    hm[(p * 4 + pr) * 128 + (w >> 2) * 64 + 16 * (w & 3) + 8 * half + i] = a;
                    // ... of the transposed product: m = column w, k = row i
                    ht[(p * 4 + pr) * 128 + (i >> 2) * 64 + 16 * (i & 3) + 8 * half + w] = a;
                }
    std::vector<double> ref(2 * NX, 0.0);
    for (size_t t = 0; t < npass * 8; ++t)
        for (int i = 0; i < 8; ++i)
            for (int w = 0; w < 8; ++w) {
                const double a = T[t * 64 + i * 8 + w];
                ref[hr[t] + i] += a * hx[hc[t] + w];
                ref[NX + hc[t] + w] += a * hx[hr[t] + i];
            }
    double *dv, *dm, *dt, *dx, *dy; uint32_t *dc, *dr;
    CK(hipMalloc(&dv, hv.size() * 8)); CK(hipMalloc(&dm, hm.size() * 8)); CK(hipMalloc(&dt, ht.size() * 8));
    CK(hipMalloc(&dx, NX * 8)); CK(hipMalloc(&dy, 2 * NX * 8)); CK(hipMalloc(&dc, hc.size() * 4)); CK(hipMalloc(&dr, hr.size() * 4));
    CK(hipMemcpy(dv, hv.data(), hv.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dm, hm.data(), hm.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, ht.data(), ht.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dx, hx.data(), NX * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dc, hc.data(), hc.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dr, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
    const unsigned blocks = 4096;
    for (int variant = 0; variant < 6; ++variant) {
        const char *names[] = {"valu_row", "mfma_row", "valu_sym", "mfma_sym", "mfma_row(map1)", "mfma_sym(map1)"};
        auto launch = [&]() {
            if (variant == 0) valu_kernel<false><<<blocks, 256>>>(dv, dc, dr, dx, dy, npass);
            if (variant == 1) mfma_kernel<false, 0><<<blocks, 256>>>(dm, dt, dc, dr, dx, dy, npass);
            if (variant == 2) valu_kernel<true><<<blocks, 256>>>(dv, dc, dr, dx, dy, npass);
            if (variant == 3) mfma_kernel<true, 0><<<blocks, 256>>>(dm, dt, dc, dr, dx, dy, npass);
            if (variant == 4) mfma_kernel<false, 1><<<blocks, 256>>>(dm, dt, dc, dr, dx, dy, npass);
            if (variant == 5) mfma_kernel<true, 1><<<blocks, 256>>>(dm, dt, dc, dr, dx, dy, npass);
        };
        CK(hipMemset(dy, 0, 2 * NX * 8));
        launch();
        CK(hipDeviceSynchronize());
        std::vector<double> got(2 * NX);
        CK(hipMemcpy(got.data(), dy, 2 * NX * 8, hipMemcpyDeviceToHost));
        double err = 0.0;
        const size_t upto = (variant == 2 || variant == 3 || variant == 5) ? 2 * NX : NX;
        for (size_t q = 0; q < upto; ++q) err = fmax(err, fabs(got[q] - ref[q]) / (1.0 + fabs(ref[q])));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        for (int r = 0; r < 20; ++r) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-9s %zu tiles (%.0f MB of values): %.2f us per sweep, %.2f TB/s of values, max rel err %.1e\n", names[variant],
               npass * 8, npass * 4096 / 1e6, ms * 1e3 / 20, npass * 4096 / (ms / 20 * 1e-3) / 1e12, err);
    }
    return 0;
}
