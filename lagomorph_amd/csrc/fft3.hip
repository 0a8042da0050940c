// FluidMetric sharp/flat as three LDS-tiled passes (float32, 3D, extents 2^a, 3*2^a, 5*2^a), gfx950.
// The transforms and the per-frequency operator are in fft_lds.hpp (host-verifiable phase
// functions); this file holds the kernels around them and the launch logic.
//
//   zy_forward_kernel : grid = nn*3*nx planes, 512 threads, LDS = ny*(nz/2+1)*8 B (planes below 80 KB: two per CU);
//                       zy_forward_persist_kernel: one persistent 1024-thread workgroup per CU for larger planes
//   fluid_xpass2_kernel: grid = nn*(ny*nz/32 + ny/16) tiles of 3 x nx x 16 bins, 256 threads;
//                       fluid_xpass2_persist_kernel: two persistent workgroups per CU once the launch is large
//   zy_inverse_kernel : grid = nn*3*nx planes (and its persistent form)
//
// HBM traffic per call: 6 passes of 4 B/voxel-component (read m, write+read+write+read the
// spectrum, write out) + the coefficient table, against 14 for rocFFT's 3D plan + operator.
#include <algorithm>
#include "fft3_sizes.hpp"

namespace lago {

template <int NY, int NZ>
using ZYK = fl::ZY<typename SzOf<NY>::T, typename SzOf<NZ / 2>::T>;
// the fused 2D kernel holds BOTH component planes: above 80 KB of LDS it has its CU to itself and runs 1024 threads
template <int NY, int NZ>
using ZYK2D = fl::ZY<typename SzOf<NY>::T, typename SzOf<NZ / 2>::T, (2 * NY * (NZ / 2 + 1) * 8 > 80 * 1024 ? 1024 : 512)>;

template <int NY, int NZ>
__global__ __launch_bounds__((ZYK<NY, NZ>::THREADS)) void zy_forward_kernel(fl::ZYArgs a) {
    using K = ZYK<NY, NZ>;
    extern __shared__ __align__(16) unsigned char lago_smem[];
    float2 *P = reinterpret_cast<float2 *>(lago_smem), *tw = P + K::NY * K::PZ;
    const size_t p = a.rev ? a.total - 1u - blockIdx.x : blockIdx.x;
    const float *in = a.in + p * (size_t)(K::NY * K::NZ);
    float2 *mainp = a.main_ + p * (size_t)(K::NY * K::NZH), *nyqp = a.nyq + p * (size_t)K::NY;
#pragma unroll
    for (int ph = 0; ph < K::NPH; ++ph) {
        K::fwd_phase(ph, threadIdx.x, in, mainp, nyqp, P, tw);
        if (ph + 1 < K::NPH) __syncthreads();
    }
}

template <int NY, int NZ>
__global__ __launch_bounds__((ZYK<NY, NZ>::THREADS)) void zy_inverse_kernel(fl::ZYArgs a) {
    using K = ZYK<NY, NZ>;
    extern __shared__ __align__(16) unsigned char lago_smem[];
    float2 *P = reinterpret_cast<float2 *>(lago_smem), *tw = P + K::NY * K::PZ;
    const size_t p = a.rev ? a.total - 1u - blockIdx.x : blockIdx.x;
    float *out = a.out + p * (size_t)(K::NY * K::NZ);
    const float2 *mainp = a.main_ + p * (size_t)(K::NY * K::NZH), *nyqp = a.nyq + p * (size_t)K::NY;
#pragma unroll
    for (int ph = 0; ph < K::NPH_INV; ++ph) {
        K::inv_phase(ph, threadIdx.x, out, mainp, nyqp, P, tw, a.oscale);
        if (ph + 1 < K::NPH_INV) __syncthreads();
    }
}


template <int NY, int NZ>
__global__ __launch_bounds__((ZYK<NY, NZ>::THREADS)) void zy_forward_persist_kernel(fl::ZYArgs a) {
    using K = ZYK<NY, NZ>;
    extern __shared__ __align__(16) unsigned char lago_smem[];
    float2 *P = reinterpret_cast<float2 *>(lago_smem), *tw = P + K::NY * K::PZ;
    K::fill_twiddles(threadIdx.x, tw);
    float4 v[K::KV];
    auto at = [&](size_t q) { return a.rev ? (size_t)a.total - 1 - q : q; };   // launch direction (common.hpp)
    size_t pq = blockIdx.x;
    K::fwd_load(threadIdx.x, a.in + at(pq) * (size_t)(K::NY * K::NZ), v);
    for (; pq < a.total; pq += gridDim.x) {
        const size_t p = at(pq);
        K::fwd_fill(threadIdx.x, v, P);
        __syncthreads();
        const bool more = pq + gridDim.x < a.total;
        const float4 *nin = reinterpret_cast<const float4 *>(a.in + (more ? at(pq + gridDim.x) : p) * (size_t)(K::NY * K::NZ));
        float2 *mainp = a.main_ + p * (size_t)(K::NY * K::NZH), *nyqp = a.nyq + p * (size_t)K::NY;
#pragma unroll
        for (int ph = 1; ph < K::NPH; ++ph) {
            constexpr int NS = K::NPH - 2;   // phases the loads are spread over
            if (ph <= NS && more) {
#pragma unroll
                for (int k = (ph - 1) * K::KV / NS; k < ph * K::KV / NS; ++k)
                    if (threadIdx.x + k * K::THREADS < K::F4) v[k] = nin[threadIdx.x + k * K::THREADS];
            }
            if (ph == K::NPH - 1) {
#pragma unroll
                for (int k = 0; k < K::KV; ++k) settle(v[k]);
            }
            K::fwd_phase(ph, threadIdx.x, nullptr, mainp, nyqp, P, tw);
            __syncthreads();
        }
    }
}

template <int NY, int NZ>
__global__ __launch_bounds__((ZYK<NY, NZ>::THREADS)) void zy_inverse_persist_kernel(fl::ZYArgs a) {
    using K = ZYK<NY, NZ>;
    extern __shared__ __align__(16) unsigned char lago_smem[];
    float2 *P = reinterpret_cast<float2 *>(lago_smem), *tw = P + K::NY * K::PZ;
    K::fill_twiddles(threadIdx.x, tw);
    float4 v[K::KVX];
    auto at = [&](size_t q) { return a.rev ? (size_t)a.total - 1 - q : q; };   // launch direction (common.hpp)
    size_t pq = blockIdx.x;
    K::inv_load(threadIdx.x, a.main_ + at(pq) * (size_t)(K::NY * K::NZH), a.nyq + at(pq) * (size_t)K::NY, v);
    for (; pq < a.total; pq += gridDim.x) {
        const size_t p = at(pq);
        K::inv_fill(threadIdx.x, v, P);
        __syncthreads();
        const bool more = pq + gridDim.x < a.total;
        const size_t pn = more ? at(pq + gridDim.x) : p;
        const float2 *nmain = a.main_ + pn * (size_t)(K::NY * K::NZH), *nnyq = a.nyq + pn * (size_t)K::NY;
        float *out = a.out + p * (size_t)(K::NY * K::NZ);
#pragma unroll
        for (int ph = 1; ph < K::NPH_INV; ++ph) {
            constexpr int NS = K::NPH_INV - 2;
            if (ph <= NS && more) {
#pragma unroll
                for (int k = (ph - 1) * K::KV / NS; k < ph * K::KV / NS; ++k)
                    if (threadIdx.x + k * K::THREADS < K::F4)
                        v[k] = reinterpret_cast<const float4 *>(nmain)[threadIdx.x + k * K::THREADS];
                if (ph == NS) K::inv_load_c0(threadIdx.x, nmain, nnyq, v);   // after the plane's own use of that slot
            }
            if (ph == K::NPH_INV - 1) {
#pragma unroll
                for (int k = 0; k < K::KVX; ++k) settle(v[k]);
            }
            K::inv_phase(ph, threadIdx.x, out, nullptr, nullptr, P, tw, a.oscale);
            __syncthreads();
        }
    }
}


// ---- 2D fields: the whole FluidMetricOperator.forward (metric.py:11-19) of one batch item in ONE kernel ----------
// Both component planes of an (H, W) field fit the LDS together up to about 128 x 128 (2 x 66.5 KB): real 2D
// transform of both (the zy phases above, H along "y", W along "z"), the reference's 2 x 2 operator per frequency
// (fluid_kernel_2d, cuda/metric.cu:162-218: the expressions and roundings of metric.hip, coefficients straight from
// the cos / sin LUTs), inverse transform, store: 8 bytes of HBM traffic per value instead of the 28 of
// rocFFT R2C + operator kernel + rocFFT C2R (and three launches less).
template <typename R>
__device__ __forceinline__ R fl2d_safe_sqrt(R x) {  // cuda/metric.cu:14-18
    if ((double)x < 1e-8) return (R)1e-4;
    return (R)sqrtf((float)x);
}

template <int NY, int NZ, bool INV>
__global__ __launch_bounds__((ZYK2D<NY, NZ>::THREADS)) void fluid2d_kernel(float *__restrict__ out, const float *__restrict__ m,
                                                                       const float *__restrict__ cosX, const float *__restrict__ sinX,
                                                                       const float *__restrict__ cosY, const float *__restrict__ sinY,
                                                                       double alpha, double beta, double gamma, float scale,
                                                                       float oscale) {
    using K = ZYK2D<NY, NZ>;
    using SY = typename SzOf<NY>::T;
    using SZH = typename SzOf<NZ / 2>::T;
    constexpr int NT = K::THREADS, PZ = K::PZ, NZH = K::NZH;
    extern __shared__ __align__(16) unsigned char lago_smem[];
    float2 *P0 = reinterpret_cast<float2 *>(lago_smem), *P1 = P0 + NY * PZ, *tw = P1 + NY * PZ;
    const size_t n = blockIdx.x;
    const float *in0 = m + n * 2 * (size_t)(NY * NZ), *in1 = in0 + (size_t)(NY * NZ);
    float *out0 = out + n * 2 * (size_t)(NY * NZ), *out1 = out0 + (size_t)(NY * NZ);
    // forward transform of both planes, up to and including the unpack phase (the store phase is not needed)
#pragma unroll
    for (int ph = 0; ph < K::NPH - 1; ++ph) {
        K::fwd_phase(ph, threadIdx.x, in0, nullptr, nullptr, P0, tw);
        K::fwd_phase(ph, threadIdx.x, in1, nullptr, nullptr, P1, tw);
        __syncthreads();
    }
    // operator: bin (kx, ky) of the half spectrum sits at row pos_of(kx), column pos_of(ky) (ky = NZH: the spare column)
    for (int w = threadIdx.x; w < NY * (NZH + 1); w += NT) {
        const int kx = w / (NZH + 1), ky = w - kx * (NZH + 1);
        const int at = fl::pos_of<SY>(kx) * PZ + (ky == NZH ? NZH : fl::pos_of<SZH>(ky));
        const float wx = cosX[kx], wy = cosY[ky];
        const float lambda = (float)__builtin_fma(alpha, (double)(wx + wy), gamma);
        const float l00 = (float)__builtin_fma(-beta, (double)wx, (double)lambda);
        const float l11 = (float)__builtin_fma(-beta, (double)wy, (double)lambda);
        const float l10 = (float)(beta * (double)sinX[kx] * (double)sinY[ky]);
        const float L00 = lg_fma(l00, l00, l10 * l10);
        const float L10 = lg_fma(l00, l10, l10 * l11);
        const float L11 = lg_fma(l11, l11, l10 * l10);
        float ooG00 = 0, G10 = 0, ooG11 = 0;
        if (INV) {  // cuda/metric.cu:20-45
            ooG00 = (float)(1. / (double)fl2d_safe_sqrt(L00));
            G10 = L10 * ooG00;
            ooG11 = lg_fma(-G10, G10, L11);
            ooG11 = (float)(1. / (double)fl2d_safe_sqrt(ooG11));
        }
        const float2 a = P0[at], b = P1[at];
        float X[2] = {a.x, a.y}, Y[2] = {b.x, b.y};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float bX = X[q], bY = Y[q];
            if (INV) {  // cuda/metric.cu:80-101
                const float y0 = bX * ooG00;
                const float y1 = lg_fma(-G10, y0, bY) * ooG11;
                bY = y1 * ooG11;
                bX = lg_fma(-G10, bY, y0) * ooG00;
            } else {  // cuda/metric.cu:132-143
                const float x = lg_fma(L00, bX, L10 * bY);
                bY = lg_fma(L10, bX, L11 * bY);
                bX = x;
            }
            X[q] = bX * scale; Y[q] = bY * scale;
        }
        P0[at] = make_float2(X[0], X[1]);
        P1[at] = make_float2(Y[0], Y[1]);
    }
    __syncthreads();
    // what inv_fill does to column 0: FA + i FB (FB = the Nyquist column), so that the inverse y transform returns
    // (X[0](y), X[NZH](y)) in one complex column
    for (int kx = threadIdx.x; kx < 2 * NY; kx += NT) {
        float2 *row = (kx < NY ? P0 : P1) + fl::pos_of<SY>(kx < NY ? kx : kx - NY) * PZ;
        const float2 fa = row[0], fb = row[NZH];
        row[0] = make_float2(fa.x - fb.y, fa.y + fb.x);
    }
    __syncthreads();
#pragma unroll
    for (int ph = 1; ph < K::NPH_INV; ++ph) {
        K::inv_phase(ph, threadIdx.x, out0, nullptr, nullptr, P0, tw, oscale);
        K::inv_phase(ph, threadIdx.x, out1, nullptr, nullptr, P1, tw, oscale);
        if (ph + 1 < K::NPH_INV) __syncthreads();
    }
}

// (H, W) the fused 2D kernel is instantiated for: both planes + the twiddle table within 160 KB of LDS
#define LAGO_2D_SHAPES(X)                                                                                   \
    X(64, 64) X(64, 96) X(64, 128) X(96, 64) X(96, 96) X(96, 128) X(128, 64) X(128, 96) X(128, 128)        \
    X(32, 64) X(32, 128) X(160, 64) X(160, 96) X(192, 64) X(192, 96) X(256, 64) X(64, 160) X(96, 160)       \
    X(64, 192) X(96, 192) X(64, 256)

bool fluid2d_supported(int64_t h, int64_t w) {
    bool ok = false;
#define X(H, W) ok = ok || (h == H && w == W);
    LAGO_2D_SHAPES(X)
#undef X
    return ok;
}

template <int NY, int NZ>
static hipError_t fluid2d_launch(float *out, const float *m, int inverse, const float *cosX, const float *sinX,
                                 const float *cosY, const float *sinY, double alpha, double beta, double gamma,
                                 int64_t nn, hipStream_t s, float oscale) {
    using K = ZYK2D<NY, NZ>;
    constexpr size_t smem = (size_t)(2 * NY * K::PZ + K::TWN) * sizeof(float2);
    static_assert(smem <= 160 * 1024, "two planes do not fit the LDS");
    const float scale = (float)(1.0 / ((double)NY * (double)NZ));
    if (inverse) {
        auto k = fluid2d_kernel<NY, NZ, true>;
        hipError_t e = allow_smem(k, smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3((uint32_t)nn), dim3(K::THREADS), smem, s, out, m, cosX, sinX, cosY, sinY, alpha, beta, gamma, scale, oscale);
    } else {
        auto k = fluid2d_kernel<NY, NZ, false>;
        hipError_t e = allow_smem(k, smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3((uint32_t)nn), dim3(K::THREADS), smem, s, out, m, cosX, sinX, cosY, sinY, alpha, beta, gamma, scale, oscale);
    }
    return hipSuccess;
}

int fluid_metric_2d(float *out, const float *m, int inverse, const float *cosX, const float *sinX, const float *cosY,
                    const float *sinY, double alpha, double beta, double gamma, int64_t nn, int64_t h, int64_t w,
                    hipStream_t s, float oscale) {
    hipError_t e = hipErrorInvalidValue;
#define X(H, W) \
    if (h == H && w == W) e = fluid2d_launch<H, W>(out, m, inverse, cosX, sinX, cosY, sinY, alpha, beta, gamma, nn, s, oscale);
    LAGO_2D_SHAPES(X)
#undef X
    if (e != hipSuccess) return fail_hip(e, "fluid_metric (2D)");
    return finish_launch(s, "fluid_metric");
}

// ---- host side ---------------------------------------------------------------------------------

// (ny, nz) planes the zy passes are instantiated for: every pair of {64, 96, 128, 160, 192} (160^3 is the volume of
// BASELINE configs[4]), plus the power-of-two shapes of round 1 with ny = 32 / 256 or nz = 256.  The plane
// ny * (nz/2 + 1) complex must fit the 160 KB of LDS together with the lcm(ny, nz)-entry twiddle table.
#define LAGO_ZY_SHAPES(X)                                                                              \
    X(64, 64) X(64, 96) X(64, 128) X(64, 160) X(64, 192) X(96, 64) X(96, 96) X(96, 128) X(96, 160) X(96, 192)      \
    X(128, 64) X(128, 96) X(128, 128) X(128, 160) X(128, 192) X(160, 64) X(160, 96) X(160, 128) X(160, 160) X(160, 192) \
    X(192, 64) X(192, 96) X(192, 128) X(192, 160) X(192, 192)                                          \
    X(32, 64) X(32, 128) X(32, 256) X(64, 256) X(128, 256) X(256, 64) X(256, 128)                      \
    X(208, 176) X(176, 176) X(176, 208)                                                                \
    X(112, 96) X(96, 112) X(112, 112) X(128, 112) X(112, 128) X(224, 160) X(160, 224) X(224, 128)      \
    X(144, 144) X(176, 144) X(144, 176) X(240, 160) X(160, 240)                                        \
    X(104, 88) X(88, 88) X(88, 104) X(120, 120) X(80, 80)


static bool zy_instantiated(int64_t ny, int64_t nz) {
    bool ok = false;
#define X(NY, NZ) ok = ok || (ny == NY && nz == NZ);
    LAGO_ZY_SHAPES(X)
#undef X
    return ok;
}
static bool big_plane_supported(int64_t ny, int64_t nz) { return big_sizes_instantiated(ny, nz) && !zy_instantiated(ny, nz); }

bool fluid_native_supported(int64_t nx, int64_t ny, int64_t nz) {
    bool okx = false, okyz = big_plane_supported(ny, nz);
#define X(N) okx = okx || nx == N;
    LAGO_X_SIZES(X)
#undef X
#define X(NY, NZ) okyz = okyz || (ny == NY && nz == NZ);
    LAGO_ZY_SHAPES(X)
#undef X
    // a Nyquist plane whose rows are not whole tiles (ny % 16 = 8) needs the x-pass instantiations that mask the short tile:
    // those of the lengths 88, 104, 120 (fft_lds.hpp: XPass::TAIL)
    if (ny % 16 != 0 && nx % 16 == 0) return false;
    return okx && okyz;
}

std::atomic<int> g_zy_persist{1};  // 1: persistent prefetching zy kernels for planes above 80 KB of LDS

template <int NY, int NZ>
static hipError_t zy_launch(const fl::ZYArgs &a, bool inverse, hipStream_t s) {
    using K = ZYK<NY, NZ>;
    static_assert(K::SMEM <= 160 * 1024, "plane does not fit the LDS");
    // one workgroup per CU (plane above 80 KB) and the next plane's registers fit beside the transform's (not the
    // 256-point rows / columns, which spill): persistent grid with register prefetch
    constexpr bool kPersist = K::SMEM > 80 * 1024 && K::KV <= 9 && NY < 256 && NZ < 256 && !(NY == 128 && NZ == 192);   // (128, 192: six full slots + the column-0 slot: 16 spilled registers)
    if constexpr (kPersist) if (g_zy_persist) {
        const uint32_t grid = std::min<uint32_t>(a.total, 256u);
        if (inverse) {
            auto k = zy_inverse_persist_kernel<NY, NZ>;
            hipError_t e = allow_smem(k, K::SMEM);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(k, dim3(grid), dim3(K::THREADS), K::SMEM, s, a);
        } else {
            auto k = zy_forward_persist_kernel<NY, NZ>;
            hipError_t e = allow_smem(k, K::SMEM);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(k, dim3(grid), dim3(K::THREADS), K::SMEM, s, a);
        }
        return hipSuccess;
    }
    if (inverse) {
        auto k = zy_inverse_kernel<NY, NZ>;
        hipError_t e = allow_smem(k, K::SMEM);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(a.total), dim3(K::THREADS), K::SMEM, s, a);
    } else {
        auto k = zy_forward_kernel<NY, NZ>;
        hipError_t e = allow_smem(k, K::SMEM);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(a.total), dim3(K::THREADS), K::SMEM, s, a);
    }
    return hipSuccess;
}

static hipError_t zy_dispatch(int64_t ny, int64_t nz, const fl::ZYArgs &a, bool inverse, hipStream_t s) {
#define X(NY, NZ) \
    if (ny == NY && nz == NZ) return zy_launch<NY, NZ>(a, inverse, s);
    LAGO_ZY_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}

std::atomic<int> g_xpass_ipw{0};  // batch items per x-pass workgroup; 0 = by the size of the launch (below)
#ifdef LAGO_PROFILING  // profiling builds only (results are wrong when != 7): bit 0 zy forward, bit 1 x pass, bit 2 zy inverse
std::atomic<int> g_native_stage_mask{7};
#define LAGO_STAGE_MASK ((int)g_native_stage_mask)
#else
#define LAGO_STAGE_MASK 7
#endif

// out = irfftn(operator(rfftn(m))) * scale.  tab: split-layout coefficient table (fluid_coef_launch
// with split = 1).  work: nn*3*nx*ny*(nz/2+1) complex.
int fluid_metric_native(float *out, const float *m, float *work, const float *tab, int inverse, int64_t nn,
                        int64_t nx, int64_t ny, int64_t nz, double scale, hipStream_t s, float oscale) {
    const int64_t nzh = nz / 2, planes = nn * 3 * nx;
    const int64_t items = ny * nzh / 16 + (ny + 15) / 16;   // tiles of 16 consecutive (r, q) positions + the Nyquist plane's (the last one may be short)
    if (planes >= (1ll << 31) || nn * items >= (1ll << 31)) return fail_invalid("fluid_metric: batch too large");
    fl::ZYArgs za;
    za.in = m;
    za.out = out;
    za.main_ = reinterpret_cast<float2 *>(work);
    za.nyq = za.main_ + (size_t)planes * ny * nzh;
    za.total = (uint32_t)planes;
    za.oscale = oscale;   // (read by the inverse zy pass only)
    fl::XArgs xa;
    xa.main_ = za.main_;
    xa.nyq = za.nyq;
    xa.tabM = tab;
    xa.tabN = tab + (size_t)nx * ny * nzh * 6;
    xa.ny = (int)ny;
    xa.nzh = (int)nzh;
    xa.nch = (int)(ny * nzh / 16);
    xa.items_per_n = (int)items;
    xa.scale = (float)scale;
    xa.nn = (int)nn;
    int ipw = g_xpass_ipw;
    const int stages = LAGO_STAGE_MASK;
    if (ipw <= 0) {
        // two batch items per workgroup halve the reads of the coefficient table (24 B per bin), but only a launch
        // with several rounds of workgroups to spare can afford workgroups of twice the length: small per-GPU
        // batches (strong scaling: 4 items per GPU at 8 ranks) keep one item per workgroup
        const int64_t slots = 256 * std::max<int64_t>(1, (160 * 1024) / (3 * nx * 17 * 8 + 2048));
        ipw = (nn / 2) * items >= 8 * slots ? 2 : 1;   // (measured: tools/ab_fluid.py)
    }
    xa.ipw = ipw;
    xa.total = (uint32_t)((nn + xa.ipw - 1) / xa.ipw * items);
    hipError_t e = hipSuccess;
    za.rev = next_direction();
    const bool big = !zy_instantiated(ny, nz);   // (fluid_native_supported: then the rows + columns route covers the plane)
    if (big && (nn * nx * ((nz / 2 + 15) / 16) >= (1ll << 31) || planes * ny / 64 >= (1ll << 31))) return fail_invalid("fluid_metric: batch too large");
    if (stages & 1) e = big ? big_zy_dispatch(nx, ny, nz, nn, za, false, s) : zy_dispatch(ny, nz, za, false, s);
    if (e != hipSuccess) return fail_hip(e, "fluid_metric (zy forward)");
    xa.rev = next_direction();
    if (stages & 2) e = xpass2_dispatch(nx, xa, inverse != 0, s);
    if (e != hipSuccess) return fail_hip(e, "fluid_metric (x pass)");
    za.rev = next_direction();
    if (stages & 4) e = big ? big_zy_dispatch(nx, ny, nz, nn, za, true, s) : zy_dispatch(ny, nz, za, true, s);
    if (e != hipSuccess) return fail_hip(e, "fluid_metric (zy inverse)");
    return finish_launch(s, "fluid_metric");
}

}  // namespace lago

#ifdef LAGO_PROFILING
extern "C" void lago_debug_fluid_stage_mask(int m) { lago::g_native_stage_mask = m; }
#endif
// tuning settings (speed only; include/lagomorph_hip.h: lago_tuning)
namespace lago {
void tune_fluid_passes(int ipw, int zy_persist, int xpass_wide, int xpass_persist) {
    g_xpass_ipw = ipw;
    g_zy_persist = zy_persist;
    g_xpass_wide = xpass_wide;
    g_xpass_persist = xpass_persist;
}
}  // namespace lago
