// Shared by fft3.hip (zy passes, 2D kernel, rows + columns route, host logic) and fft3x.hip (the x pass): the length ->
// factorisation table, the x lengths the passes are instantiated for, and two helpers of the persistent kernels.
#pragma once
#include <atomic>
#include "common.hpp"
#include "fft_lds.hpp"

namespace lago {

// transform length -> its factorisation R * 2^L2 (fft_lds.hpp)
template <int N> struct SzOf;
template <> struct SzOf<32> { using T = fl::Sz<1, 5>; };
template <> struct SzOf<48> { using T = fl::Sz<3, 4>; };
template <> struct SzOf<64> { using T = fl::Sz<1, 6>; };
template <> struct SzOf<80> { using T = fl::Sz<5, 4>; };
template <> struct SzOf<96> { using T = fl::Sz<3, 5>; };
template <> struct SzOf<128> { using T = fl::Sz<1, 7>; };
template <> struct SzOf<160> { using T = fl::Sz<5, 5>; };
template <> struct SzOf<192> { using T = fl::Sz<3, 6>; };
template <> struct SzOf<256> { using T = fl::Sz<1, 8>; };
// 176 = 11 * 16 and 208 = 13 * 16 (half lengths 88, 104): the 176 x 208 x 176 volumes of the OASIS brain images (round 6)
template <> struct SzOf<88> { using T = fl::Sz<11, 3>; };
template <> struct SzOf<104> { using T = fl::Sz<13, 3>; };
template <> struct SzOf<176> { using T = fl::Sz<11, 4>; };
template <> struct SzOf<208> { using T = fl::Sz<13, 4>; };
// 7, 9 and 15 times a power of two: 96 x 112 x 96 and 192 x 224 x 160 (cropped MNI grids), 144^3 / 144 x 176 x 144, 240
template <> struct SzOf<56> { using T = fl::Sz<7, 3>; };
template <> struct SzOf<112> { using T = fl::Sz<7, 4>; };
template <> struct SzOf<224> { using T = fl::Sz<7, 5>; };
template <> struct SzOf<72> { using T = fl::Sz<9, 3>; };
template <> struct SzOf<144> { using T = fl::Sz<9, 4>; };
template <> struct SzOf<120> { using T = fl::Sz<15, 3>; };
template <> struct SzOf<240> { using T = fl::Sz<15, 4>; };
// 8 times an odd factor (ny % 16 = 8: the Nyquist plane's last tile is half a tile) and the half lengths that go with them:
// 88 x 104 x 88 (the 176 x 208 x 176 grid at half resolution: multiscale momenta), 120^3, 80^3 (160^3 at half resolution)
template <> struct SzOf<40> { using T = fl::Sz<5, 3>; };
template <> struct SzOf<44> { using T = fl::Sz<11, 2>; };
template <> struct SzOf<52> { using T = fl::Sz<13, 2>; };
template <> struct SzOf<60> { using T = fl::Sz<15, 2>; };


// Planes whose LDS image leaves room for ONE workgroup per CU only (160 x 160: 104 KB) have nobody to hide their
// global loads behind: a grid of one persistent workgroup per CU walks the planes and requests plane p + grid into
// registers (7 float4 per thread) while it transforms plane p.  Two details of that loop, both measured
// (tools/probes/zy_probe.hip, profiles/r03_zy_passes.md):
//  * the loads are spread over the transform's phases (one or two per phase) instead of issued together, so that their
//    issue time hides behind other waves' LDS work;
//  * the prefetched registers are waited for BEFORE the store phase (`settle`: an empty asm that takes them as
//    operands).  hipcc counts loads and stores in one counter and waits vmcnt(0) at the next fill, i.e. for this
//    plane's stores to be acknowledged; with the wait in front of the stores the fill finds nothing pending.
__device__ __forceinline__ void settle(float4 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

template <typename Kern>
static hipError_t allow_smem(Kern k, size_t smem) {
    if (smem <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)smem);
}

// x lengths of the x pass (fft3x.hip)
#define LAGO_X_SIZES(X) X(64) X(96) X(128) X(160) X(192) X(256) X(176) X(208) X(112) X(224) X(144) X(240) X(88) X(104) X(120) X(80)

// fft3x.hip
hipError_t xpass2_dispatch(int64_t nx, const fl::XArgs &a, bool inverse, hipStream_t s);
extern std::atomic<int> g_xpass_persist, g_xpass_wide;

// fft3b.hip: the rows + columns route
bool big_sizes_instantiated(int64_t ny, int64_t nz);
hipError_t big_zy_dispatch(int64_t nx, int64_t ny, int64_t nz, int64_t nn, const fl::ZYArgs &za, bool inverse, hipStream_t s);

}  // namespace lago
