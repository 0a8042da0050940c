// Probe: what bounds the persistent zy passes at 160 x 160 (one 104 KB plane per CU)?  The library's own phase
// functions (fft_lds.hpp) in the library's persistent loop, with parts switched off:
//   mode 0  the kernel as shipped
//   mode 1  memory only: load -> fill -> store phase (no transform stages)
//   mode 2  LDS only: fill from registers, all stages, no global loads after the first and no global stores
//   mode 3  the kernel as shipped, one workgroup per plane (not persistent)
//   mode 4  shipped + the prefetched registers are waited for BEFORE the store phase (an empty asm that takes them as
//           operands), so that the fill of the next plane does not wait behind this plane's stores (hipcc emits
//           vmcnt(0) at the fill: loads and stores count together)
//   mode 5  mode 4 + the prefetch loads spread over the stage phases (one per phase) instead of issued together
// usage: zy_probe [batch = 8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../lagomorph_amd/csrc/fft_lds.hpp"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

using namespace lago;
using K1024 = fl::ZY<fl::Sz<5, 5>, fl::Sz<5, 4>>;   // 160 x 160, 1024 threads
using K512 = fl::ZY<fl::Sz<5, 5>, fl::Sz<5, 4>, 512>;
using K = K1024;

template <int MODE, class K = K1024>
__global__ __launch_bounds__(K::THREADS) void zy_fwd(fl::ZYArgs a, int never) {
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *P = reinterpret_cast<float2 *>(smem), *tw = P + K::NY * K::PZ;
    K::fill_twiddles(threadIdx.x, tw);
    float4 v[K::KV];
    size_t pq = blockIdx.x;
    K::fwd_load(threadIdx.x, a.in + pq * (size_t)(K::NY * K::NZ), v);
    for (; pq < a.total; pq += gridDim.x) {
        const size_t p = pq;
        K::fwd_fill(threadIdx.x, v, P);
        __syncthreads();
        const bool more = pq + gridDim.x < a.total;
        const float4 *nin = reinterpret_cast<const float4 *>(a.in + (pq + gridDim.x) * (size_t)(K::NY * K::NZ));
        if (MODE != 5 && (MODE != 2 || never))
            if (more) K::fwd_load(threadIdx.x, a.in + (pq + gridDim.x) * (size_t)(K::NY * K::NZ), v);
        float2 *mainp = a.main_ + p * (size_t)(K::NY * K::NZH), *nyqp = a.nyq + p * (size_t)K::NY;
#pragma unroll
        for (int ph = 1; ph < K::NPH; ++ph) {
            if (MODE == 1 && ph < K::NPH - 1) continue;
            if (MODE == 2 && ph == K::NPH - 1 && !never) continue;
            if (MODE == 5 && more && ph < K::NPH - 1) {
                constexpr int NS = K::NPH - 2;
#pragma unroll
                for (int k = (ph - 1) * K::KV / NS; k < ph * K::KV / NS; ++k)
                    if (threadIdx.x + k * K::THREADS < K::F4) v[k] = nin[threadIdx.x + k * K::THREADS];
            }
            if (MODE >= 4 && ph == K::NPH - 1) {
#pragma unroll
                for (int k = 0; k < K::KV; ++k) asm volatile("" : "+v"(v[k].x), "+v"(v[k].y), "+v"(v[k].z), "+v"(v[k].w));
            }
            K::fwd_phase(ph, threadIdx.x, nullptr, mainp, nyqp, P, tw);
            __syncthreads();
        }
    }
}

template <int MODE, class K = K1024>
__global__ __launch_bounds__(K::THREADS) void zy_inv(fl::ZYArgs a, int never) {
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *P = reinterpret_cast<float2 *>(smem), *tw = P + K::NY * K::PZ;
    K::fill_twiddles(threadIdx.x, tw);
    float4 v[K::KVX];
    size_t pq = blockIdx.x;
    K::inv_load(threadIdx.x, a.main_ + pq * (size_t)(K::NY * K::NZH), a.nyq + pq * (size_t)K::NY, v);
    for (; pq < a.total; pq += gridDim.x) {
        const size_t p = pq;
        K::inv_fill(threadIdx.x, v, P);
        __syncthreads();
        const bool more = pq + gridDim.x < a.total;
        const float2 *nmain = a.main_ + (pq + gridDim.x) * (size_t)(K::NY * K::NZH), *nnyq = a.nyq + (pq + gridDim.x) * (size_t)K::NY;
        if (MODE != 5 && (MODE != 2 || never))
            if (more) {
                const size_t pn = pq + gridDim.x;
                K::inv_load(threadIdx.x, a.main_ + pn * (size_t)(K::NY * K::NZH), a.nyq + pn * (size_t)K::NY, v);
            }
        float *out = a.out + p * (size_t)(K::NY * K::NZ);
#pragma unroll
        for (int ph = 1; ph < K::NPH_INV; ++ph) {
            if (MODE == 1 && ph < K::NPH_INV - 1) continue;
            if (MODE == 2 && ph == K::NPH_INV - 1 && !never) continue;
            if (MODE == 5 && more && ph < K::NPH_INV - 1) {
                constexpr int NS = K::NPH_INV - 2;
#pragma unroll
                for (int k = (ph - 1) * K::KV / NS; k < ph * K::KV / NS; ++k)
                    if (threadIdx.x + k * K::THREADS < K::F4) v[k] = reinterpret_cast<const float4 *>(nmain)[threadIdx.x + k * K::THREADS];
                if (ph == NS) K::inv_load_c0(threadIdx.x, nmain, nnyq, v);
            }
            if (MODE >= 4 && ph == K::NPH_INV - 1) {
#pragma unroll
                for (int k = 0; k < K::KVX; ++k) asm volatile("" : "+v"(v[k].x), "+v"(v[k].y), "+v"(v[k].z), "+v"(v[k].w));
            }
            K::inv_phase(ph, threadIdx.x, out, nullptr, nullptr, P, tw);
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(K::THREADS) void zy_fwd_plain(fl::ZYArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *P = reinterpret_cast<float2 *>(smem), *tw = P + K::NY * K::PZ;
    const size_t p = blockIdx.x;
#pragma unroll
    for (int ph = 0; ph < K::NPH; ++ph) {
        K::fwd_phase(ph, threadIdx.x, a.in + p * (size_t)(K::NY * K::NZ), a.main_ + p * (size_t)(K::NY * K::NZH),
                     a.nyq + p * (size_t)K::NY, P, tw);
        if (ph + 1 < K::NPH) __syncthreads();
    }
}

// 128 x 128 planes, one workgroup per plane (two per CU), the second workgroup of every CU optionally started late:
// does breaking the lock-step of a launch with few rounds help?  (usage: zy_probe <batch> 128)
using K128 = fl::ZY<fl::Sz<1, 7>, fl::Sz<1, 6>>;
__global__ __launch_bounds__(K128::THREADS) void zy128_fwd_plain(fl::ZYArgs a, int sleeps, int every) {
    using K = K128;
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *P = reinterpret_cast<float2 *>(smem), *tw = P + K::NY * K::PZ;
    if ((blockIdx.x / every) & 1)
        for (int i = 0; i < sleeps; ++i) __builtin_amdgcn_s_sleep(127);
    const size_t p = blockIdx.x;
#pragma unroll
    for (int ph = 0; ph < K::NPH; ++ph) {
        K::fwd_phase(ph, threadIdx.x, a.in + p * (size_t)(K::NY * K::NZ), a.main_ + p * (size_t)(K::NY * K::NZH),
                     a.nyq + p * (size_t)K::NY, P, tw);
        if (ph + 1 < K::NPH) __syncthreads();
    }
}

template <typename F>
static float time_us(F launch, int iters = 20) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
}

static int main128(int nn) {
    using K = K128;
    const uint32_t planes = (uint32_t)nn * 3 * 128;
    const size_t nreal = (size_t)planes * K::NY * K::NZ, nc = (size_t)planes * K::NY * (K::NZH + 1);
    float *in; float2 *work;
    CK(hipMalloc(&in, nreal * 4)); CK(hipMalloc(&work, nc * 8));
    CK(hipMemset(in, 0, nreal * 4)); CK(hipMemset(work, 0, nc * 8));
    fl::ZYArgs a;
    a.in = in; a.out = nullptr; a.main_ = work; a.nyq = work + (size_t)planes * K::NY * K::NZH; a.total = planes; a.rev = 0;
    const double bytes = (double)nreal * 4 + (double)nc * 8;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(zy128_fwd_plain), hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::SMEM));
    printf("128 x 128 planes: %u (batch %d), %d threads, LDS %zu B\n", planes, nn, K::THREADS, (size_t)K::SMEM);
    for (int every : {256, 1})
        for (int sleeps : {0, 1, 2, 3, 4}) {
            float t = time_us([&] { hipLaunchKernelGGL(zy128_fwd_plain, dim3(planes), dim3(K::THREADS), K::SMEM, 0, a, sleeps, every); });
            printf("late start of every other %s: %d x s_sleep(127)  %7.1f us  %5.2f TB/s\n", every == 1 ? "workgroup      " : "256 workgroups ", sleeps, t, bytes / (t * 1e-6) / 1e12);
        }
    return 0;
}

int main(int argc, char **argv) {
    const int nn = argc > 1 ? atoi(argv[1]) : 8;
    if (argc > 2 && atoi(argv[2]) == 128) return main128(nn);
    const uint32_t planes = (uint32_t)nn * 3 * 160;
    const size_t nreal = (size_t)planes * K::NY * K::NZ, nc = (size_t)planes * K::NY * (K::NZH + 1);
    float *in, *out; float2 *work;
    CK(hipMalloc(&in, nreal * 4)); CK(hipMalloc(&out, nreal * 4)); CK(hipMalloc(&work, nc * 8));
    CK(hipMemset(in, 0, nreal * 4)); CK(hipMemset(work, 0, nc * 8));
    fl::ZYArgs a;
    a.in = in; a.out = out; a.main_ = work; a.nyq = work + (size_t)planes * K::NY * K::NZH; a.total = planes; a.rev = 0;
    const double bytes = (double)nreal * 4 + (double)nc * 8;
#define ALLOW(k) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::SMEM))
    ALLOW(zy_fwd<0>); ALLOW(zy_fwd<1>); ALLOW(zy_fwd<2>); ALLOW(zy_inv<0>); ALLOW(zy_inv<1>); ALLOW(zy_inv<2>); ALLOW(zy_fwd_plain);
    ALLOW(zy_fwd<4>); ALLOW(zy_fwd<5>); ALLOW(zy_inv<4>); ALLOW(zy_inv<5>);
    ALLOW((zy_fwd<4, K512>)); ALLOW((zy_fwd<5, K512>)); ALLOW((zy_inv<4, K512>)); ALLOW((zy_inv<5, K512>)); ALLOW((zy_fwd<2, K512>));
    printf("160 x 160 planes: %u (batch %d), %d phases forward, %d inverse, LDS %zu B, %.0f MB per pass\n", planes, nn, K::NPH,
           K::NPH_INV, (size_t)K::SMEM, bytes / 1e6);
    const uint32_t grid = planes < 256 ? planes : 256;
    float t;
#define RUN(name, k)                                                                              \
    t = time_us([&] { hipLaunchKernelGGL(k, dim3(grid), dim3(K::THREADS), K::SMEM, 0, a, 0); }); \
    printf("%-44s %7.1f us  %5.2f TB/s\n", name, t, bytes / (t * 1e-6) / 1e12);
    RUN("forward, shipped persistent", zy_fwd<0>)
    RUN("forward, memory only (load, fill, store)", zy_fwd<1>)
    RUN("forward, LDS only (fill + stages)", zy_fwd<2>)
    RUN("forward, waits before the stores", zy_fwd<4>)
    RUN("forward, + loads spread over the phases", zy_fwd<5>)
    RUN("inverse, shipped persistent", zy_inv<0>)
    RUN("inverse, waits before the stores", zy_inv<4>)
    RUN("inverse, + loads spread over the phases", zy_inv<5>)
    RUN("inverse, memory only", zy_inv<1>)
    RUN("inverse, LDS only", zy_inv<2>)
#define RUN5(name, k)                                                                      \
    t = time_us([&] { hipLaunchKernelGGL((k), dim3(grid), dim3(512), K::SMEM, 0, a, 0); }); \
    printf("%-44s %7.1f us  %5.2f TB/s\n", name, t, bytes / (t * 1e-6) / 1e12);
    RUN5("512 thr: forward, LDS only", (zy_fwd<2, K512>))
    RUN5("512 thr: forward, waits before the stores", (zy_fwd<4, K512>))
    RUN5("512 thr: forward, + loads spread", (zy_fwd<5, K512>))
    RUN5("512 thr: inverse, waits before the stores", (zy_inv<4, K512>))
    RUN5("512 thr: inverse, + loads spread", (zy_inv<5, K512>))
    t = time_us([&] { hipLaunchKernelGGL(zy_fwd_plain, dim3(planes), dim3(K::THREADS), K::SMEM, 0, a); });
    printf("%-44s %7.1f us  %5.2f TB/s\n", "forward, one workgroup per plane", t, bytes / (t * 1e-6) / 1e12);
    return 0;
}
