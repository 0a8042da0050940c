// Planes without a one-kernel zy instantiation (above the LDS: 256 x 256, 192 x 224 ...; or simply not listed): the zy
// transform as ROWS + COLUMNS (fft_lds.hpp: ZRows, YPass) around the x pass of fft3x.hip.  A translation unit of its own so
// that the three halves of the tuned passes compile in parallel.
#include <algorithm>
#include "fft3_sizes.hpp"

namespace lago {

// lengths the rows + columns route is instantiated for: every (ny, nz) pair of them that has no one-kernel zy
// instantiation -- the planes above the LDS (256 x 256, 224 x 224, 192 x 224, 208 x 192, 240 x 224 ...) and mixed planes
// nobody listed (176 x 160, 144 x 128, 208 x 208 ...).  (nz / 2 = 72, 88, 104, 120: the column pass masks the half tile at the
// end of every row.)
// With every multiple of 16 from 64 to 256 in both lists (their odd factors are 1 ... 15) and in the x lengths, ANY float32
// volume whose three extents are such multiples runs LDS-tiled passes: three launches where its plane has a one-kernel
// instantiation, five otherwise.
#define LAGO_BIG_Y_SIZES(X) X(64) X(80) X(96) X(112) X(128) X(144) X(160) X(176) X(192) X(208) X(224) X(240) X(256)
#define LAGO_BIG_Z_SIZES(X) X(64) X(80) X(96) X(112) X(128) X(144) X(160) X(176) X(192) X(208) X(224) X(240) X(256)

// ---- planes above the LDS: rows + columns (fft_lds.hpp: ZRows, YPass) ---------------------------------------------
template <int NZ>
using ZRK = fl::ZRows<typename SzOf<NZ / 2>::T>;

template <int NZ>
__global__ __launch_bounds__(512) void zrows_forward_kernel(const float *__restrict__ in, float2 *__restrict__ main_, uint32_t total, int rev) {
    using R = ZRK<NZ>;
    using K = typename R::K;
    extern __shared__ __align__(16) unsigned char lago_smem[];
    float2 *P = reinterpret_cast<float2 *>(lago_smem), *tw = P + R::RB * R::PZ;
    const size_t blk = rev ? total - 1u - blockIdx.x : blockIdx.x;
    const float *inp = in + blk * (size_t)(R::RB * NZ);
#pragma unroll
    for (int ph = 0; ph <= R::GZ + 1; ++ph) {
        K::fwd_phase(ph, threadIdx.x, inp, nullptr, nullptr, P, tw);
        __syncthreads();
    }
    R::fwd_store(threadIdx.x, P, main_ + blk * (size_t)(R::RB * R::NZH));
}

template <int NZ>
__global__ __launch_bounds__(512) void zrows_inverse_kernel(float *__restrict__ out, const float2 *__restrict__ main_, uint32_t total, int rev,
                                                            float oscale) {
    using R = ZRK<NZ>;
    using K = typename R::K;
    extern __shared__ __align__(16) unsigned char lago_smem[];
    float2 *P = reinterpret_cast<float2 *>(lago_smem), *tw = P + R::RB * R::PZ;
    const size_t blk = rev ? total - 1u - blockIdx.x : blockIdx.x;
    K::fill_twiddles(threadIdx.x, tw);
    R::inv_fill(threadIdx.x, main_ + blk * (size_t)(R::RB * R::NZH), P);
    __syncthreads();
    float *outp = out + blk * (size_t)(R::RB * NZ);
#pragma unroll
    for (int ph = R::GYK + 1; ph <= R::GYK + R::GZ + 2; ++ph) {
        K::inv_phase(ph, threadIdx.x, outp, nullptr, nullptr, P, tw, oscale);
        if (ph < R::GYK + R::GZ + 2) __syncthreads();
    }
}

template <int NY> constexpr int ypass_threads() { return NY >= 208 ? 512 : 256; }   // (tiles above 80 KB are alone on their CU)

template <int NY, bool FWD>
__global__ __launch_bounds__((ypass_threads<NY>())) void ypass_kernel(fl::YArgs a) {
    using K = fl::YPass<typename SzOf<NY>::T, ypass_threads<NY>()>;
    extern __shared__ __align__(16) unsigned char lago_smem[];
    float2 *buf = reinterpret_cast<float2 *>(lago_smem), *tw = buf + 3 * K::NY * K::KCP;
    const typename K::Block b = K::locate(a, block_order(blockIdx.x, a.total, a.rev));
#pragma unroll
    for (int ph = 0; ph < K::NPH; ++ph) {
        if (FWD) K::fwd_phase(ph, threadIdx.x, b, buf, tw);
        else K::inv_phase(ph, threadIdx.x, b, buf, tw);
        if (ph + 1 < K::NPH) __syncthreads();
    }
}

template <int NZ>
static hipError_t zrows_launch(const fl::ZYArgs &a, uint32_t blocks, bool inverse, hipStream_t s) {
    using R = ZRK<NZ>;
    if (inverse) {
        auto k = zrows_inverse_kernel<NZ>;
        hipError_t e = allow_smem(k, R::SMEM);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(R::NT), R::SMEM, s, a.out, a.main_, blocks, a.rev, a.oscale);
    } else {
        auto k = zrows_forward_kernel<NZ>;
        hipError_t e = allow_smem(k, R::SMEM);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(R::NT), R::SMEM, s, a.in, a.main_, blocks, a.rev);
    }
    return hipSuccess;
}
template <int NY>
static hipError_t ypass_launch(const fl::YArgs &a, bool inverse, hipStream_t s) {
    using K = fl::YPass<typename SzOf<NY>::T, ypass_threads<NY>()>;
    if (inverse) {
        auto k = ypass_kernel<NY, false>;
        hipError_t e = allow_smem(k, K::SMEM);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(a.total), dim3(K::NT), K::SMEM, s, a);
    } else {
        auto k = ypass_kernel<NY, true>;
        hipError_t e = allow_smem(k, K::SMEM);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(a.total), dim3(K::NT), K::SMEM, s, a);
    }
    return hipSuccess;
}
// the zy transform of planes above the LDS: rows then columns (forward), columns then rows (inverse)
hipError_t big_zy_dispatch(int64_t nx, int64_t ny, int64_t nz, int64_t nn, const fl::ZYArgs &za, bool inverse, hipStream_t s) {
    fl::YArgs ya;
    ya.main_ = za.main_; ya.nyq = za.nyq; ya.nx = (int)nx; ya.ny = (int)ny; ya.nzh = (int)(nz / 2); ya.ntile = (int)((nz / 2 + 15) / 16);
    ya.total = (uint32_t)(nn * nx * ya.ntile); ya.rev = za.rev;
    const uint32_t blocks = (uint32_t)((uint64_t)za.total * (uint64_t)ny / 64u);
    hipError_t e = hipErrorInvalidValue;
    auto rows = [&]() {
        hipError_t r = hipErrorInvalidValue;
#define X(N) if (nz == N) r = zrows_launch<N>(za, blocks, inverse, s);
        LAGO_BIG_Z_SIZES(X)
#undef X
        return r;
    };
    auto cols = [&]() {
        hipError_t r = hipErrorInvalidValue;
#define X(N) if (ny == N) r = ypass_launch<N>(ya, inverse, s);
        LAGO_BIG_Y_SIZES(X)
#undef X
        return r;
    };
    if (!inverse) {
        e = rows();
        if (e == hipSuccess) e = cols();
    } else {
        e = cols();
        if (e == hipSuccess) e = rows();
    }
    return e;
}

bool big_sizes_instantiated(int64_t ny, int64_t nz) {
    bool oky = false, okz = false;
#define X(N) oky = oky || ny == N;
    LAGO_BIG_Y_SIZES(X)
#undef X
#define X(N) okz = okz || nz == N;
    LAGO_BIG_Z_SIZES(X)
#undef X
    return oky && okz;
}

}  // namespace lago
