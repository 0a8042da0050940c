// The x pass of the fluid metric's LDS-tiled FFT passes (fft_lds.hpp: XPass): 3 components x nx x 16 frequency bins per
// workgroup -- FFT along x, the per-frequency operator from the cached coefficient table, inverse FFT, in place.  Split off
// fft3.hip in round 6 (its ~90 instantiations are most of that file's compile time: the two now build in parallel).
#include <algorithm>
#include "fft3_sizes.hpp"

namespace lago {

// threads per x-pass workgroup: 256, except for the 256-point tile (104 KB: one workgroup per CU, which 512 threads
// serve 12 % faster).  Measured (tools/ab_fluid.py): wider workgroups LOSE 3-14 % at 128, 160 and 192 points, where
// two or three 256-thread workgroups share a CU and their 120+ VGPRs per thread would cost the second one.
template <int NX> constexpr int xpass_wide() { return NX >= 208 ? 512 : 256; }   // (208, 224, 240: 86 - 100 KB, alone on their CU as well)

template <int NX, bool INV, int NT>
__global__ __launch_bounds__(NT) void fluid_xpass2_kernel(fl::XArgs a) {
    using K = fl::XPass<typename SzOf<NX>::T, INV, NT>;
    extern __shared__ __align__(16) unsigned char lago_smem[];
    float2 *buf = reinterpret_cast<float2 *>(lago_smem), *tw = buf + 3 * K::NX * K::KCP;
    const uint32_t blk = block_order(blockIdx.x, a.total, a.rev);
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

// The x pass as a persistent grid (two workgroups per CU): each workgroup walks a contiguous run of (bin tile, batch
// item) pairs -- batch items innermost, so the coefficients stay in registers until the bin tile changes -- and
// requests the next tile into registers, spread over the phases of the current one; the prefetched registers are
// settled before the store phase (see zy_forward_persist_kernel).  One-shot workgroups, two per CU, overlap their
// load / compute / store only by chance: 388 -> ... us at 32 x 3 x 128^3 (tools/ab_fluid.py).
template <int NX, bool INV, int NT>
__global__ __launch_bounds__(NT) void fluid_xpass2_persist_kernel(fl::XArgs a) {
    using K = fl::XPass<typename SzOf<NX>::T, INV, NT>;
    extern __shared__ __align__(16) unsigned char lago_smem[];
    float2 *buf = reinterpret_cast<float2 *>(lago_smem), *tw = buf + 3 * K::NX * K::KCP;
    const uint32_t T = (uint32_t)a.nn * (uint32_t)a.items_per_n;
    const uint32_t q0 = (uint32_t)((uint64_t)blockIdx.x * T / gridDim.x), q1 = (uint32_t)((uint64_t)(blockIdx.x + 1) * T / gridDim.x);
    if (q0 >= q1) return;
    auto at = [&](uint32_t q) {   // pair number q of the launch: bin tile q / nn, batch item q % nn
        const uint32_t qq = a.rev ? T - 1u - q : q;
        return K::locate(a, qq % (uint32_t)a.nn, qq / (uint32_t)a.nn);
    };
    K::fill_twiddles(threadIdx.x, tw);
    typename K::Regs r;
    float4 v[K::KLD];
    typename K::Block b = at(q0);
    const float *tb_held = nullptr;
#pragma unroll
    for (int k = 0; k < K::KLD; ++k) K::load_one(threadIdx.x, b, v, k);
    for (uint32_t q = q0; q < q1; ++q) {
        K::fill(threadIdx.x, v, buf);
        if (b.tb != tb_held) {   // workgroup-uniform: a new bin tile
            K::load_coef(threadIdx.x, r, b);
            tb_held = b.tb;
        }
        __syncthreads();
        const bool more = q + 1 < q1;
        const typename K::Block bn = more ? at(q + 1) : b;
#pragma unroll
        for (int ph = 1; ph < K::NPH; ++ph) {
            constexpr int NS = K::NPH - 2;   // phases the loads are spread over
            if (ph <= NS && more) {
#pragma unroll
                for (int k = (ph - 1) * K::KLD / NS; k < ph * K::KLD / NS; ++k) K::load_one(threadIdx.x, bn, v, k);
            }
            if (ph == K::NPH - 1) {
#pragma unroll
                for (int k = 0; k < K::KLD; ++k) settle(v[k]);
            }
            K::phase(ph, threadIdx.x, r, b, buf, tw, a.scale, false);
            __syncthreads();
        }
        b = bn;
    }
}


std::atomic<int> g_xpass_persist{1};  // 1: persistent x-pass grid (two workgroups per CU) once the launch has enough pairs; 2: always

template <int NX, int NT>
static hipError_t xpass2_launch_nt(const fl::XArgs &a, bool inverse, hipStream_t s) {
    using K0 = fl::XPass<typename SzOf<NX>::T, false, NT>;
    // persistent: tiles that fit a CU twice, and at least eight (bin tile, batch item) pairs per workgroup (below that the
    // one-shot workgroups are as fast or faster: 142 against 146 us per sharp at 4 x 128^3, tools/ab_fluid.py)
    const uint32_t per_cu = (uint32_t)std::min<size_t>(2, (160 * 1024) / K0::SMEM);
    const uint64_t pairs = (uint64_t)a.nn * (uint64_t)a.items_per_n;
    const int mode = g_xpass_persist;   // 2 (tests): whatever the size of the launch
    const uint32_t grid = (uint32_t)std::min<uint64_t>(256u * per_cu, pairs);
    // (192 points: 263 VGPRs, one 256-thread workgroup per CU -- stays with the one-shot workgroups)
    // (the persistent kernels exist only for the lengths whose tile fits a CU twice)
    constexpr bool kCanPersist = K0::SMEM * 2 <= 160 * 1024 && (NX <= 160 || NX == 176);
    const bool persist = mode && kCanPersist && (mode >= 2 || pairs >= 8ull * grid) && pairs < (1ull << 32);
    if (inverse) {
        using K = fl::XPass<typename SzOf<NX>::T, true, NT>;
        if constexpr (kCanPersist) if (persist) {
            auto k = fluid_xpass2_persist_kernel<NX, true, NT>;
            hipError_t e = allow_smem(k, K::SMEM);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(k, dim3(grid), dim3(NT), K::SMEM, s, a);
            return hipSuccess;
        }
        auto k = fluid_xpass2_kernel<NX, true, NT>;
        hipError_t e = allow_smem(k, K::SMEM);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(a.total), dim3(NT), K::SMEM, s, a);
    } else {
        using K = fl::XPass<typename SzOf<NX>::T, false, NT>;
        if constexpr (kCanPersist) if (persist) {
            auto k = fluid_xpass2_persist_kernel<NX, false, NT>;
            hipError_t e = allow_smem(k, K::SMEM);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(k, dim3(grid), dim3(NT), K::SMEM, s, a);
            return hipSuccess;
        }
        auto k = fluid_xpass2_kernel<NX, false, NT>;
        hipError_t e = allow_smem(k, K::SMEM);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(a.total), dim3(NT), K::SMEM, s, a);
    }
    return hipSuccess;
}

std::atomic<int> g_xpass_wide{1};

template <int NX>
static hipError_t xpass2_launch(const fl::XArgs &a, bool inverse, hipStream_t s) {
    if constexpr (xpass_wide<NX>() != 256)
        if (g_xpass_wide) return xpass2_launch_nt<NX, xpass_wide<NX>()>(a, inverse, s);
    return xpass2_launch_nt<NX, 256>(a, inverse, s);
}

hipError_t xpass2_dispatch(int64_t nx, const fl::XArgs &a, bool inverse, hipStream_t s) {
#define X(N) \
    if (nx == N) return xpass2_launch<N>(a, inverse, s);
    LAGO_X_SIZES(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace lago
