// FluidMetric sharp/flat as three LDS-tiled passes (float32, 3D, power-of-two extents), gfx950.
// The transforms and the per-frequency operator are in fft_lds.hpp (host-verifiable phase
// functions); this file holds the kernels around them and the launch logic.
//
//   zy_forward_kernel : grid = nn*3*nx planes, 512 threads, LDS = ny*(nz/2+1)*8 B
//   fluid_xpass2_kernel: grid = nn*(ny*(nz/32) + ny/16) tiles of 3 x nx x 16 bins, 256 threads
//   zy_inverse_kernel : grid = nn*3*nx planes
//
// HBM traffic per call: 6 passes of 4 B/voxel-component (read m, write+read+write+read the
// spectrum, write out) + the coefficient table, against 14 for rocFFT's 3D plan + operator.
#include "common.hpp"
#include "fft_lds.hpp"

namespace lago {

// threads per plane: 1024 once a plane has at least 2048 float4 (two per thread)
constexpr int zy_threads(int logny, int lognz) { return logny + lognz >= 13 ? 1024 : 512; }

template <int LOGNY, int LOGNZ>
__global__ __launch_bounds__(zy_threads(LOGNY, LOGNZ)) void zy_forward_kernel(fl::ZYArgs a) {
    using K = fl::ZY<LOGNY, LOGNZ, zy_threads(LOGNY, LOGNZ)>;
    extern __shared__ __align__(16) unsigned char lago_smem[];
    float2 *P = reinterpret_cast<float2 *>(lago_smem), *tw = P + K::NY * K::PZ;
    const size_t p = blockIdx.x;
    const float *in = a.in + p * (size_t)(K::NY * K::NZ);
    float2 *mainp = a.main_ + p * (size_t)(K::NY * K::NZH), *nyqp = a.nyq + p * (size_t)K::NY;
#pragma unroll
    for (int ph = 0; ph < K::NPH; ++ph) {
        K::fwd_phase(ph, threadIdx.x, in, mainp, nyqp, P, tw);
        if (ph + 1 < K::NPH) __syncthreads();
    }
}

template <int LOGNY, int LOGNZ>
__global__ __launch_bounds__(zy_threads(LOGNY, LOGNZ)) void zy_inverse_kernel(fl::ZYArgs a) {
    using K = fl::ZY<LOGNY, LOGNZ, zy_threads(LOGNY, LOGNZ)>;
    extern __shared__ __align__(16) unsigned char lago_smem[];
    float2 *P = reinterpret_cast<float2 *>(lago_smem), *tw = P + K::NY * K::PZ;
    const size_t p = blockIdx.x;
    float *out = a.out + p * (size_t)(K::NY * K::NZ);
    const float2 *mainp = a.main_ + p * (size_t)(K::NY * K::NZH), *nyqp = a.nyq + p * (size_t)K::NY;
#pragma unroll
    for (int ph = 0; ph < K::NPH_INV; ++ph) {
        K::inv_phase(ph, threadIdx.x, out, mainp, nyqp, P, tw);
        if (ph + 1 < K::NPH_INV) __syncthreads();
    }
}

template <int LOGNX, bool INV>
__global__ __launch_bounds__(256) void fluid_xpass2_kernel(fl::XArgs a) {
    using K = fl::XPass<LOGNX, INV, 256>;
    extern __shared__ __align__(16) unsigned char lago_smem[];
    float2 *buf = reinterpret_cast<float2 *>(lago_smem), *tw = buf + 3 * K::NX * K::KCP;
    const uint32_t blk = xcd_swizzle(blockIdx.x, a.total);
    typename K::Block b = K::locate(a, blk);
    typename K::Regs r;
    // consecutive batch items under the same coefficients (held in registers): the 24-byte table
    // entry of a bin is then read once per `ipw` items instead of once per item
    const uint32_t n0 = blk / (uint32_t)a.items_per_n * (uint32_t)a.ipw;
    const int nit = min(a.ipw, a.nn - (int)n0);
    for (int it = 0; it < nit; ++it) {
#pragma unroll
        for (int ph = 0; ph < K::NPH; ++ph) {
            K::phase(ph, threadIdx.x, r, b, buf, tw, a.scale, it == 0);
            if (ph + 1 < K::NPH || it + 1 < nit) __syncthreads();
        }
        b.base += (size_t)3 * K::NX * b.xs;  // next batch item, same bins
    }
}

// ---- host side ---------------------------------------------------------------------------------

static int ilog2(int64_t v) {
    int l = 0;
    while ((1ll << l) < v) ++l;
    return (1ll << l) == v ? l : -1;
}

bool fluid_native_supported(int64_t nx, int64_t ny, int64_t nz) {
    const int lx = ilog2(nx), ly = ilog2(ny), lz = ilog2(nz);
    return lx >= 6 && lx <= 8 && ly >= 5 && ly <= 8 && lz >= 6 && lz <= 8 && ly + lz <= 15;
}

template <typename Kern>
static hipError_t allow_smem(Kern k, size_t smem) {
    if (smem <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)smem);
}

template <int LY, int LZ>
static hipError_t zy_launch(const fl::ZYArgs &a, bool inverse, hipStream_t s) {
    constexpr int NT = zy_threads(LY, LZ);
    using K = fl::ZY<LY, LZ, NT>;
    if (inverse) {
        auto k = zy_inverse_kernel<LY, LZ>;
        hipError_t e = allow_smem(k, K::SMEM);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(a.total), dim3(NT), K::SMEM, s, a);
    } else {
        auto k = zy_forward_kernel<LY, LZ>;
        hipError_t e = allow_smem(k, K::SMEM);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(a.total), dim3(NT), K::SMEM, s, a);
    }
    return hipSuccess;
}

template <int LY>
static hipError_t zy_by_z(int lz, const fl::ZYArgs &a, bool inverse, hipStream_t s) {
    if (lz == 6) return zy_launch<LY, 6>(a, inverse, s);
    if (lz == 7) return zy_launch<LY, 7>(a, inverse, s);
    if constexpr (LY <= 7) {
        if (lz == 8) return zy_launch<LY, 8>(a, inverse, s);
    }
    return hipErrorInvalidValue;
}

static hipError_t zy_dispatch(int ly, int lz, const fl::ZYArgs &a, bool inverse, hipStream_t s) {
    switch (ly) {
        case 5: return zy_by_z<5>(lz, a, inverse, s);
        case 6: return zy_by_z<6>(lz, a, inverse, s);
        case 7: return zy_by_z<7>(lz, a, inverse, s);
        case 8: return zy_by_z<8>(lz, a, inverse, s);
    }
    return hipErrorInvalidValue;
}

template <int LX>
static hipError_t xpass2_launch(const fl::XArgs &a, bool inverse, hipStream_t s) {
    if (inverse) {
        using K = fl::XPass<LX, true, 256>;
        auto k = fluid_xpass2_kernel<LX, true>;
        hipError_t e = allow_smem(k, K::SMEM);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(a.total), dim3(256), K::SMEM, s, a);
    } else {
        using K = fl::XPass<LX, false, 256>;
        auto k = fluid_xpass2_kernel<LX, false>;
        hipError_t e = allow_smem(k, K::SMEM);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(a.total), dim3(256), K::SMEM, s, a);
    }
    return hipSuccess;
}

int g_xpass_ipw = 2;  // batch items per x-pass workgroup
int g_native_stage_mask = 7;  // profiling only: bit 0 zy forward, bit 1 x pass, bit 2 zy inverse

// out = irfftn(operator(rfftn(m))) * scale.  tab: split-layout coefficient table (fluid_coef_launch
// with split = 1).  work: nn*3*nx*ny*(nz/2+1) complex.
int fluid_metric_native(float *out, const float *m, float *work, const float *tab, int inverse, int64_t nn,
                        int64_t nx, int64_t ny, int64_t nz, double scale, hipStream_t s) {
    const int lx = ilog2(nx), ly = ilog2(ny), lz = ilog2(nz);
    const int64_t nzh = nz / 2, planes = nn * 3 * nx;
    const int64_t items = ny * (nzh / 16) + ny / 16;
    if (planes >= (1ll << 31) || nn * items >= (1ll << 31)) return fail_invalid("fluid_metric: batch too large");
    fl::ZYArgs za;
    za.in = m;
    za.out = out;
    za.main_ = reinterpret_cast<float2 *>(work);
    za.nyq = za.main_ + (size_t)planes * ny * nzh;
    za.total = (uint32_t)planes;
    fl::XArgs xa;
    xa.main_ = za.main_;
    xa.nyq = za.nyq;
    xa.tabM = tab;
    xa.tabN = tab + (size_t)nx * ny * nzh * 6;
    xa.ny = (int)ny;
    xa.nzh = (int)nzh;
    xa.nch = (int)(nzh / 16);
    xa.items_per_n = (int)items;
    xa.scale = (float)scale;
    xa.nn = (int)nn;
    xa.ipw = g_xpass_ipw > 0 ? g_xpass_ipw : 1;
    xa.total = (uint32_t)((nn + xa.ipw - 1) / xa.ipw * items);
    hipError_t e = hipSuccess;
    if (g_native_stage_mask & 1) e = zy_dispatch(ly, lz, za, false, s);
    if (e != hipSuccess) return fail_hip(e, "fluid_metric (zy forward)");
    if (g_native_stage_mask & 2) {
        if (lx == 6) e = xpass2_launch<6>(xa, inverse != 0, s);
        else if (lx == 7) e = xpass2_launch<7>(xa, inverse != 0, s);
        else e = xpass2_launch<8>(xa, inverse != 0, s);
    }
    if (e != hipSuccess) return fail_hip(e, "fluid_metric (x pass)");
    if (g_native_stage_mask & 4) e = zy_dispatch(ly, lz, za, true, s);
    if (e != hipSuccess) return fail_hip(e, "fluid_metric (zy inverse)");
    return finish_launch(s, "fluid_metric");
}

}  // namespace lago

extern "C" void lago_debug_fluid_stage_mask(int m) { lago::g_native_stage_mask = m; }
extern "C" void lago_debug_xpass_ipw(int n) { lago::g_xpass_ipw = n; }
