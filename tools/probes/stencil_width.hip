// Probe: is a coalesced-row stencil kernel (jtv_backward: 48 dword loads per voxel, all rows streamed or L1/L2 hits)
// bound by the NUMBER of vector-memory instructions?  Same bytes, same arithmetic:
//   V1  one voxel per lane, dword loads            (the product's scheme)
//   V2  two z-adjacent voxels per lane, dwordx2 loads (rows x+-1, y+-1 and centre as pairs, z-1 / z+2 as dwords)
//   V4  four z-adjacent voxels per lane, dwordx4
// F fields of S^3 x B floats, 6-neighbour clamped stencil on each, one output field.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int S = 128, B = 8, F = 9;
__device__ __forceinline__ int cl(int x, int n) { return x < 0 ? 0 : (x > n - 1 ? n - 1 : x); }

template <int W>
__global__ __launch_bounds__(256) void k(float* out, const float* in) {
    const size_t nv = (size_t)S * S * S;
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;   // group of W voxels along z
    const size_t s0 = q * W;
    if (s0 >= nv * B) return;
    const size_t n = s0 / nv, r = s0 % nv;
    const int i = r / (S * S), j = (r / S) % S, kz = r % S;
    const int xp = i + 1 < S ? S * S : 0, xm = i > 0 ? -S * S : 0, yp = j + 1 < S ? S : 0, ym = j > 0 ? -S : 0;
    float acc[W];
#pragma unroll
    for (int e = 0; e < W; ++e) acc[e] = 0.f;
    struct alignas(4 * W) V { float e[W]; };
#pragma unroll
    for (int f = 0; f < F; ++f) {
        const float* p = in + ((size_t)f * B + n) * nv + r;
        const V c = *reinterpret_cast<const V*>(p);
        const V a = *reinterpret_cast<const V*>(p + xp), b = *reinterpret_cast<const V*>(p + xm);
        const V d = *reinterpret_cast<const V*>(p + yp), g = *reinterpret_cast<const V*>(p + ym);
        const float zm = p[kz > 0 ? -1 : 0], zp = p[kz + W < S ? W : W - 1];
#pragma unroll
        for (int e = 0; e < W; ++e) {
            const float lo = e == 0 ? zm : c.e[e - 1], hi = e == W - 1 ? zp : c.e[e + 1];
            acc[e] += (a.e[e] - b.e[e]) * c.e[e] + (d.e[e] - g.e[e]) * 0.5f + (hi - lo) * 0.25f;
        }
    }
    V o;
#pragma unroll
    for (int e = 0; e < W; ++e) o.e[e] = acc[e];
    *reinterpret_cast<V*>(out + s0) = o;
}

int main() {
    const size_t nv = (size_t)S * S * S * B;
    float *in, *out;
    hipMalloc(&in, nv * F * 4);
    hipMalloc(&out, nv * 4);
    hipMemset(in, 0, nv * F * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](auto kern, int W, const char* name) {
        const unsigned grid = (unsigned)((nv / W + 255) / 256);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, in);
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, in);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms / 20 * 1e3;
        printf("%s: %.1f us  (%.0f GB/s of the %d B/voxel compulsory traffic; %d vector loads per %d voxels)\n", name, us,
               (F + 1) * 4.0 * nv / us / 1e3, (F + 1) * 4, F * 7, W);
    };
    for (int rep = 0; rep < 2; ++rep) {
        run(k<1>, 1, "V1 dword  ");
        run(k<2>, 2, "V2 dwordx2");
        run(k<4>, 4, "V4 dwordx4");
    }
    return 0;
}
