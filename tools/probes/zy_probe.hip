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
            if ((MODE == 1 || MODE == 3 || MODE == 4 || MODE == 5) && ph < K::NPH - 1) continue;
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

// The persistent x pass of the library (fft3.hip: fluid_xpass2_persist_kernel) at 128 points with parts switched off:
// mode 0 as shipped, 1 memory only (load, fill, store), 2 LDS only (no global loads after the first, no stores),
// 3 memory only with the stores going to a second buffer (out of place), 4 memory only with every tile one contiguous
// 48 KB block (what a tile-major spectrum layout would give); 5 / 6: memory only / everything with the pairs dealt
// round-robin (workgroup w takes pairs w, w + grid, ...; chunk fastest: neighbouring workgroups read the neighbouring
// 128-byte pieces of the same rows at the same time) instead of one contiguous run per workgroup.
// usage: zy_probe <batch> x
using KX = fl::XPass<fl::Sz<1, 7>, true, 256>;
template <int MODE>
__global__ __launch_bounds__(256) void xpass_probe(fl::XArgs a, int never, float2 *second) {
    using K = KX;
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *buf = reinterpret_cast<float2 *>(smem), *tw = buf + 3 * K::NX * K::KCP;
    const uint32_t T = (uint32_t)a.nn * (uint32_t)a.items_per_n;
    constexpr bool RR = MODE >= 5;
    const uint32_t q0 = RR ? 0u : (uint32_t)((uint64_t)blockIdx.x * T / gridDim.x);
    const uint32_t q1 = RR ? (T - blockIdx.x + gridDim.x - 1) / gridDim.x : (uint32_t)((uint64_t)(blockIdx.x + 1) * T / gridDim.x);
    if (q0 >= q1) return;
    auto at = [&](uint32_t q) {
        const uint32_t pr = blockIdx.x + q * gridDim.x;   // round-robin pair number (RR)
        typename K::Block bb = RR ? K::locate(a, pr / (uint32_t)a.items_per_n, pr % (uint32_t)a.items_per_n)
                                  : K::locate(a, q % (uint32_t)a.nn, q / (uint32_t)a.nn);
        if (MODE == 4) { bb.xs = 16; bb.base = a.main_ + (size_t)q * 3 * K::NX * 16; }
        return bb;
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
        if (b.tb != tb_held) { K::load_coef(threadIdx.x, r, b); tb_held = b.tb; }
        __syncthreads();
        const bool more = q + 1 < q1;
        const typename K::Block bn = more ? at(q + 1) : b;
#pragma unroll
        for (int ph = 1; ph < K::NPH; ++ph) {
            constexpr int NS = K::NPH - 2;
            if (ph <= NS && more && (MODE != 2 || never)) {
#pragma unroll
                for (int k = (ph - 1) * K::KLD / NS; k < ph * K::KLD / NS; ++k) K::load_one(threadIdx.x, bn, v, k);
            }
            if (ph == K::NPH - 1) {
#pragma unroll
                for (int k = 0; k < K::KLD; ++k) asm volatile("" : "+v"(v[k].x), "+v"(v[k].y), "+v"(v[k].z), "+v"(v[k].w));
            }
            if ((MODE == 1 || MODE == 3 || MODE == 4 || MODE == 5) && ph < K::NPH - 1) continue;
            if (MODE == 2 && ph == K::NPH - 1 && !never) continue;
            if (MODE == 3 && ph == K::NPH - 1) {
                typename K::Block bo = b;
                bo.base = second + (b.base - a.main_);
                K::phase(ph, threadIdx.x, r, bo, buf, tw, a.scale, false);
            } else
                K::phase(ph, threadIdx.x, r, b, buf, tw, a.scale, false);
            __syncthreads();
        }
        b = bn;
    }
}

template <typename F> static float time_us(F launch, int iters);
template <typename F> static float time_us(F launch) { return time_us(launch, 20); }
static int mainx(int nn) {
    using K = KX;
    const int nx = 128, ny = 128, nzh = 64;
    const size_t nmain = (size_t)nn * 3 * nx * ny * nzh, nnyq = (size_t)nn * 3 * nx * ny;
    float2 *work; float *tab;
    CK(hipMalloc(&work, (nmain + nnyq) * 8)); CK(hipMemset(work, 0, (nmain + nnyq) * 8));
    const size_t ntab = (size_t)nx * ny * (nzh + 1) * 6;
    CK(hipMalloc(&tab, ntab * 4)); CK(hipMemset(tab, 0, ntab * 4));
    fl::XArgs a;
    a.main_ = work; a.nyq = work + nmain; a.tabM = tab; a.tabN = tab + (size_t)nx * ny * nzh * 6;
    a.ny = ny; a.nzh = nzh; a.nch = nzh / 16; a.items_per_n = ny * a.nch + ny / 16; a.nn = nn; a.ipw = 1; a.scale = 1.f;
    a.total = (uint32_t)(nn * a.items_per_n); a.rev = 0;
    const double bytes = 2.0 * (double)(nmain + nnyq) * 8;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(xpass_probe<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::SMEM));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(xpass_probe<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::SMEM));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(xpass_probe<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::SMEM));
    printf("x pass, 128 points, batch %d: %u (bin tile, batch item) pairs, %d phases, LDS %zu B, %.0f MB of spectrum traffic\n", nn,
           a.total, K::NPH, (size_t)K::SMEM, bytes / 1e6);
    float t;
    t = time_us([&] { hipLaunchKernelGGL(xpass_probe<0>, dim3(512), dim3(256), K::SMEM, 0, a, 0, nullptr); });
    printf("%-44s %7.1f us  %5.2f TB/s\n", "persistent x pass as shipped", t, bytes / (t * 1e-6) / 1e12);
    t = time_us([&] { hipLaunchKernelGGL(xpass_probe<1>, dim3(512), dim3(256), K::SMEM, 0, a, 0, nullptr); });
    printf("%-44s %7.1f us  %5.2f TB/s\n", "memory only (load, fill, store)", t, bytes / (t * 1e-6) / 1e12);
    t = time_us([&] { hipLaunchKernelGGL(xpass_probe<2>, dim3(512), dim3(256), K::SMEM, 0, a, 0, nullptr); });
    printf("%-44s %7.1f us  %5.2f TB/s\n", "LDS only (fill, stages, operator)", t, bytes / (t * 1e-6) / 1e12);
    float2 *second;
    CK(hipMalloc(&second, (nmain + nnyq) * 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(xpass_probe<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::SMEM));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(xpass_probe<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::SMEM));
    t = time_us([&] { hipLaunchKernelGGL(xpass_probe<3>, dim3(512), dim3(256), K::SMEM, 0, a, 0, second); });
    printf("%-44s %7.1f us  %5.2f TB/s\n", "memory only, out of place", t, bytes / (t * 1e-6) / 1e12);
    t = time_us([&] { hipLaunchKernelGGL(xpass_probe<4>, dim3(512), dim3(256), K::SMEM, 0, a, 0, nullptr); });
    printf("%-44s %7.1f us  %5.2f TB/s\n", "memory only, contiguous 48 KB tiles", t, bytes / (t * 1e-6) / 1e12);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(xpass_probe<5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::SMEM));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(xpass_probe<6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::SMEM));
    t = time_us([&] { hipLaunchKernelGGL(xpass_probe<5>, dim3(512), dim3(256), K::SMEM, 0, a, 0, nullptr); });
    printf("%-44s %7.1f us  %5.2f TB/s\n", "memory only, pairs dealt round-robin", t, bytes / (t * 1e-6) / 1e12);
    t = time_us([&] { hipLaunchKernelGGL(xpass_probe<6>, dim3(512), dim3(256), K::SMEM, 0, a, 0, nullptr); });
    printf("%-44s %7.1f us  %5.2f TB/s\n", "everything, pairs dealt round-robin", t, bytes / (t * 1e-6) / 1e12);
    return 0;
}

template <typename F>
static float time_us(F launch, int iters) {
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
    if (argc > 2 && argv[2][0] == 'x') return mainx(nn);
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
