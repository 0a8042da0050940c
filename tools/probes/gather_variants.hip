// Probe: trilinear gather of a 3-channel field at x + u(x) (interp_forward, C = 3: 36 B/voxel), B x 128^3, on a dumped
// smooth displacement field.  Variants of how the 8 corners are fetched; outputs are checked against variant 0.
//  V0  four 8-byte pair loads per channel at zb = clamp(fz, 0, nz-2), selects for the clamped ends (the product's scheme)
//  V1  interior fast path: unclamped offsets, no selects (border lanes: V0 path)
//  V2  eight 4-byte loads per channel
//  V3  four 4-byte loads (floor z) per channel; the ceil-z value comes from lane + 1 through DPP when that lane's
//      rows are the same and its floor is one cell further; the other lanes fetch it with four masked 4-byte loads
//  V6  no image access at all (24 B/voxel streamed)       V7  image read at the voxel's own index (coalesced, 36 B/voxel)
//  V4  as V1 with the three component planes' pair loads issued before any is used (12 in flight)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
constexpr int S = 128;
typedef __amdgpu_buffer_rsrc_t Rsrc;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ Rsrc mk(const void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000); }
__device__ __forceinline__ int flr(float x) { int r; asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ int med3(int x, int lo, int hi) { int r; asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi)); return r; }
__device__ __forceinline__ float ld1(Rsrc r, unsigned off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0)); }
__device__ __forceinline__ void ld2(Rsrc r, unsigned off, float& lo, float& hi) {
    const u32x2 p = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
    const unsigned long long q = __builtin_bit_cast(unsigned long long, p);
    lo = __builtin_bit_cast(float, (unsigned)q); hi = __builtin_bit_cast(float, (unsigned)(q >> 32));
}
__device__ __forceinline__ float lerp8(const float* c, float t, float u, float v) {
    const float omt = 1.f - t, omu = 1.f - u, omv = 1.f - v;
    return fmaf(omv, fmaf(omu, fmaf(omt, c[0], t * c[1]), u * fmaf(omt, c[3], t * c[2])),
                v * fmaf(omu, fmaf(omt, c[4], t * c[5]), u * fmaf(omt, c[7], t * c[6])));
}
template <int V, int U>
__global__ __launch_bounds__(256) void k(float* __restrict__ out, const float* __restrict__ img, const float* __restrict__ u, unsigned nblk) {
    const unsigned b = blockIdx.x;
    const unsigned q8 = nblk >> 3;
    const unsigned L = b < (q8 << 3) ? (b & 7u) * q8 + (b >> 3) : b;   // XCD-contiguous order
    const unsigned nbx = (S * S * S) / (256 * U);
    const unsigned n = L / nbx, bx = L % nbx;
    const size_t nv = (size_t)S * S * S;
    const float* un = u + (size_t)n * 3 * nv;
    const float* In = img + (size_t)n * 3 * nv;
    float* on = out + (size_t)n * 3 * nv;
    const unsigned planeB = (unsigned)nv * 4u, rowB = S * 4u, slabB = S * S * 4u;
    unsigned s[U];
    float t[U], uu[U], v[U];
    unsigned rb[U][4];
    bool fhi[U], clo[U], inner[U];
#pragma unroll
    for (int e = 0; e < U; ++e) {
        s[e] = (bx * U + e) * 256 + threadIdx.x;
        const int i = s[e] / (S * S), j = (s[e] / S) % S, kk = s[e] % S;
        const float hx = (float)i + un[s[e]], hy = (float)j + un[nv + s[e]], hz = (float)kk + un[2 * nv + s[e]];
        const int fx = flr(hx), fy = flr(hy), fz = flr(hz);
        t[e] = hx - (float)fx; uu[e] = hy - (float)fy; v[e] = hz - (float)fz;
        inner[e] = (unsigned)fx < S - 1 && (unsigned)fy < S - 1 && (unsigned)fz < S - 1;
        const int x0 = med3(fx, 0, S - 1), x1 = med3(fx + 1, 0, S - 1), y0 = med3(fy, 0, S - 1), y1 = med3(fy + 1, 0, S - 1);
        const int zb = med3(fz, 0, S - 2);
        fhi[e] = fz > S - 2; clo[e] = fz < 0;
        const unsigned ff = x0 * slabB + y0 * rowB + zb * 4u, dX = x1 != x0 ? slabB : 0u, dY = y1 != y0 ? rowB : 0u;
        rb[e][0] = ff; rb[e][1] = ff + dX; rb[e][2] = ff + dX + dY; rb[e][3] = ff + dY;
    }
    bool coh[U];
    if (V == 3) {
#pragma unroll
        for (int e = 0; e < U; ++e) {
            const unsigned n0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)rb[e][0], 0x130, 0xf, 0xf, false);
            const unsigned n2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)rb[e][2], 0x130, 0xf, 0xf, false);
            coh[e] = n0 == rb[e][0] + 4u && n2 == rb[e][2] + 4u && !fhi[e] && !clo[e];
        }
    }
    float res[3][U];
    if (V == 10) {   // V10: no gathers (coalesced read at the voxel's own index) but the full select + trilinear arithmetic
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int e = 0; e < U; ++e) {
                const float x = In[(size_t)c * nv + s[e]];
                float lo[4], hi[4], c8[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) { lo[q] = x + (float)q; hi[q] = x * (float)(q + 2); }
#pragma unroll
                for (int q = 0; q < 4; ++q) { c8[q] = fhi[e] ? hi[q] : lo[q]; c8[q + 4] = clo[e] ? lo[q] : hi[q]; }
                res[c][e] = lerp8(c8, t[e], uu[e], v[e]) + (float)rb[e][2];
            }
    } else if (V == 9) {   // V9: the gathers of V0, trivial arithmetic
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const Rsrc r = mk(In + (size_t)c * nv, planeB);
#pragma unroll
            for (int e = 0; e < U; ++e) {
                float lo[4], hi[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) ld2(r, rb[e][q], lo[q], hi[q]);
                res[c][e] = ((lo[0] + hi[0]) + (lo[1] + hi[1])) + ((lo[2] + hi[2]) + (lo[3] + hi[3])) * t[e];
            }
        }
    } else if (V == 6 || V == 7) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int e = 0; e < U; ++e) res[c][e] = (V == 7 ? In[(size_t)c * nv + s[e]] : 1.f) * t[e] + uu[e] * v[e] + (float)rb[e][2];
    } else if (V == 4) {
        float lo[3][U][4], hi[3][U][4];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const Rsrc r = mk(In + (size_t)c * nv, planeB);
#pragma unroll
            for (int e = 0; e < U; ++e)
#pragma unroll
                for (int q = 0; q < 4; ++q) ld2(r, rb[e][q], lo[c][e][q], hi[c][e][q]);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int e = 0; e < U; ++e) {
                float c8[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) { c8[q] = fhi[e] ? hi[c][e][q] : lo[c][e][q]; c8[q + 4] = clo[e] ? lo[c][e][q] : hi[c][e][q]; }
                res[c][e] = lerp8(c8, t[e], uu[e], v[e]);
            }
    } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const Rsrc r = mk(In + (size_t)c * nv, planeB);
        float lo[U][4], hi[U][4];
#pragma unroll
        for (int e = 0; e < U; ++e) {
            if (V == 0 || V == 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) ld2(r, rb[e][q], lo[e][q], hi[e][q]);
            } else if (V == 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // (the +4 offset is made opaque: written plainly, the compiler merges the two dword loads back into
                    //  one dwordx2 -- which is what round 2 measured under the name V2 without noticing)
                    unsigned o2 = rb[e][q] + 4u;
                    asm volatile("" : "+v"(o2));
                    lo[e][q] = ld1(r, rb[e][q]); hi[e][q] = ld1(r, o2);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) lo[e][q] = ld1(r, rb[e][q]);
                if (!coh[e]) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) hi[e][q] = ld1(r, rb[e][q] + 4u);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < U; ++e) {
            if (V == 3) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float nb = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, lo[e][q]), 0x130, 0xf, 0xf, false));
                    if (coh[e]) hi[e][q] = nb;
                }
            }
            float c8[8];
            if (V == 1 && inner[e]) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { c8[q] = lo[e][q]; c8[q + 4] = hi[e][q]; }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) { c8[q] = fhi[e] ? hi[e][q] : lo[e][q]; c8[q + 4] = clo[e] ? lo[e][q] : hi[e][q]; }
            }
            res[c][e] = lerp8(c8, t[e], uu[e], v[e]);
        }
    }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int e = 0; e < U; ++e) on[(size_t)c * nv + s[e]] = res[c][e];
}
// V5: the workgroup covers a compact tile of TXR x TYR rows (x, y) of 128 voxels instead of consecutive flattened
// voxels, so that the rows its gathers touch are re-used from its CU's L1 (pair loads, V0 arithmetic)
template <int TXR, int TYR, int NT>
__global__ __launch_bounds__(NT) void ktile(float* __restrict__ out, const float* __restrict__ img, const float* __restrict__ u, unsigned nblk) {
    constexpr int U = TXR * TYR * S / NT, RPP = NT / S;   // passes; rows per pass
    const unsigned b = blockIdx.x;
    const unsigned q8 = nblk >> 3;
    const unsigned L = b < (q8 << 3) ? (b & 7u) * q8 + (b >> 3) : b;
    constexpr unsigned tx = S / TXR, ty = S / TYR;
    const unsigned n = L / (tx * ty), r = L % (tx * ty), bx = r / ty, by = r % ty;
    const size_t nv = (size_t)S * S * S;
    const float* un = u + (size_t)n * 3 * nv;
    const float* In = img + (size_t)n * 3 * nv;
    float* on = out + (size_t)n * 3 * nv;
    const unsigned planeB = (unsigned)nv * 4u, rowB = S * 4u, slabB = S * S * 4u;
    const int kk = threadIdx.x % S, r0 = threadIdx.x / S;
#pragma unroll
    for (int e = 0; e < U; ++e) {
        const int rr = e * RPP + r0, a = rr / TYR, bb = rr % TYR;
        const int i = bx * TXR + a, j = by * TYR + bb;
        const unsigned s = ((unsigned)i * S + j) * S + kk;
        const float hx = (float)i + un[s], hy = (float)j + un[nv + s], hz = (float)kk + un[2 * nv + s];
        const int fx = flr(hx), fy = flr(hy), fz = flr(hz);
        const float t = hx - (float)fx, uu = hy - (float)fy, v = hz - (float)fz;
        const int x0 = med3(fx, 0, S - 1), x1 = med3(fx + 1, 0, S - 1), y0 = med3(fy, 0, S - 1), y1 = med3(fy + 1, 0, S - 1);
        const int zb = med3(fz, 0, S - 2);
        const bool fhi = fz > S - 2, clo = fz < 0;
        const unsigned ff = x0 * slabB + y0 * rowB + zb * 4u, dX = x1 != x0 ? slabB : 0u, dY = y1 != y0 ? rowB : 0u;
        const unsigned rb[4] = {ff, ff + dX, ff + dX + dY, ff + dY};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const Rsrc rs = mk(In + (size_t)c * nv, planeB);
            float lo[4], hi[4], c8[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) ld2(rs, rb[q], lo[q], hi[q]);
#pragma unroll
            for (int q = 0; q < 4; ++q) { c8[q] = fhi ? hi[q] : lo[q]; c8[q + 4] = clo ? lo[q] : hi[q]; }
            on[(size_t)c * nv + s] = lerp8(c8, t, uu, v);
        }
    }
}
// V8: LDS-staged.  The workgroup covers TXR x TYR rows of 128 voxels; per channel the window
// [(TXR + 1 + 2M) x (TYR + 1 + 2M) rows] x 128 of the image (origin: displacement probed at the tile centre) is copied
// into LDS with 16-byte loads, the 8 corners are read from LDS (two ds_read_b32 per row); a voxel whose footprint leaves the
// window or the grid interior takes the global pair loads.  One window buffer, fill / barrier / gather / barrier per channel.
template <int TXR, int TYR, int M, int NT>
__global__ __launch_bounds__(NT) void klds(float* __restrict__ out, const float* __restrict__ img, const float* __restrict__ u, unsigned nblk) {
    constexpr int U = TXR * TYR * S / NT, RPP = NT / S, WX = TXR + 1 + 2 * M, WY = TYR + 1 + 2 * M, WZ = S;
    extern __shared__ float win[];
    const unsigned b = blockIdx.x;
    const unsigned q8 = nblk >> 3;
    const unsigned L = b < (q8 << 3) ? (b & 7u) * q8 + (b >> 3) : b;
    constexpr unsigned tx = S / TXR, ty = S / TYR;
    const unsigned n = L / (tx * ty), r = L % (tx * ty), bx = r / ty, by = r % ty;
    const size_t nv = (size_t)S * S * S;
    const float* un = u + (size_t)n * 3 * nv;
    const float* In = img + (size_t)n * 3 * nv;
    float* on = out + (size_t)n * 3 * nv;
    const unsigned planeB = (unsigned)nv * 4u, rowB = S * 4u, slabB = S * S * 4u;
    const int kk = threadIdx.x % S, r0 = threadIdx.x / S;
    const int x0 = bx * TXR, y0 = by * TYR;
    const size_t sc = ((size_t)(x0 + TXR / 2) * S + (y0 + TYR / 2)) * S + S / 2;
    const int wx0 = max(0, min(x0 + (int)floorf(un[sc]) - M, S - WX)), wy0 = max(0, min(y0 + (int)floorf(un[nv + sc]) - M, S - WY));
    unsigned s[U], rb[U][4];
    int la[U];
    float t[U], uu[U], v[U];
    bool fhi[U], clo[U], inw[U];
#pragma unroll
    for (int e = 0; e < U; ++e) {
        const int rr = e * RPP + r0, a = rr / TYR, bb = rr % TYR;
        const int i = x0 + a, j = y0 + bb;
        s[e] = ((unsigned)i * S + j) * S + kk;
        const float hx = (float)i + un[s[e]], hy = (float)j + un[nv + s[e]], hz = (float)kk + un[2 * nv + s[e]];
        const int fx = flr(hx), fy = flr(hy), fz = flr(hz);
        t[e] = hx - (float)fx; uu[e] = hy - (float)fy; v[e] = hz - (float)fz;
        const int X0 = med3(fx, 0, S - 1), X1 = med3(fx + 1, 0, S - 1), Y0 = med3(fy, 0, S - 1), Y1 = med3(fy + 1, 0, S - 1);
        const int zb = med3(fz, 0, S - 2);
        fhi[e] = fz > S - 2; clo[e] = fz < 0;
        const unsigned ff = X0 * slabB + Y0 * rowB + zb * 4u, dX = X1 != X0 ? slabB : 0u, dY = Y1 != Y0 ? rowB : 0u;
        rb[e][0] = ff; rb[e][1] = ff + dX; rb[e][2] = ff + dX + dY; rb[e][3] = ff + dY;
        const unsigned lx = fx - wx0, ly = fy - wy0;
        inw[e] = lx < WX - 1 && ly < WY - 1 && (unsigned)fz < S - 1;
        la[e] = ((int)lx * WY + (int)ly) * WZ + fz;
    }
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        const float* Ic = In + (size_t)c * nv;
        // fill: WX*WY rows of 128 floats, 16 bytes per lane
        for (int f = threadIdx.x; f < WX * WY * (WZ / 4); f += NT) {
            const int row = f / (WZ / 4), z4 = f % (WZ / 4), lx = row / WY, ly = row % WY;
            reinterpret_cast<float4*>(win)[f] = reinterpret_cast<const float4*>(Ic + ((size_t)(wx0 + lx) * S + (wy0 + ly)) * S)[z4];
        }
        __syncthreads();
        const Rsrc rs = mk(Ic, planeB);
#pragma unroll
        for (int e = 0; e < U; ++e) {
            float c8[8];
            if (inw[e]) {
                const float* w = win + la[e];
                c8[0] = w[0]; c8[4] = w[1]; c8[1] = w[WY * WZ]; c8[5] = w[WY * WZ + 1];
                c8[2] = w[WY * WZ + WZ]; c8[6] = w[WY * WZ + WZ + 1]; c8[3] = w[WZ]; c8[7] = w[WZ + 1];
            } else {
                float lo[4], hi[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) ld2(rs, rb[e][q], lo[q], hi[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) { c8[q] = fhi[e] ? hi[q] : lo[q]; c8[q + 4] = clo[e] ? lo[q] : hi[q]; }
            }
            on[(size_t)c * nv + s[e]] = lerp8(c8, t[e], uu[e], v[e]);
        }
        __syncthreads();
    }
}
template <int TXR, int TYR, int M, int NT> float runl(float* out, const float* img, const float* u, int B) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const unsigned nblk = (unsigned)((size_t)B * (S / TXR) * (S / TYR));
    const size_t smem = (size_t)(TXR + 1 + 2 * M) * (TYR + 1 + 2 * M) * S * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(klds<TXR, TYR, M, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    for (int i = 0; i < 2; ++i) klds<TXR, TYR, M, NT><<<nblk, NT, smem>>>(out, img, u, nblk);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < 10; ++i) klds<TXR, TYR, M, NT><<<nblk, NT, smem>>>(out, img, u, nblk);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return 1e3f * ms / 10;
}
template <int TXR, int TYR, int NT> float runt(float* out, const float* img, const float* u, int B) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const unsigned nblk = (unsigned)((size_t)B * (S / TXR) * (S / TYR));
    for (int i = 0; i < 2; ++i) ktile<TXR, TYR, NT><<<nblk, NT>>>(out, img, u, nblk);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < 10; ++i) ktile<TXR, TYR, NT><<<nblk, NT>>>(out, img, u, nblk);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return 1e3f * ms / 10;
}
template <int V, int U> float run(float* out, const float* img, const float* u, int B) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const unsigned nblk = (unsigned)((size_t)B * S * S * S / (256 * U));
    for (int i = 0; i < 2; ++i) k<V, U><<<nblk, 256>>>(out, img, u, nblk);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < 10; ++i) k<V, U><<<nblk, 256>>>(out, img, u, nblk);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return 1e3f * ms / 10;
}
int main(int argc, char** argv) {
    const int B = argc > 2 ? atoi(argv[2]) : 32;
    const size_t nv = (size_t)S * S * S, n3 = (size_t)B * 3 * nv;
    std::vector<float> hu(n3), hv(n3);
    FILE* f = argc > 1 ? fopen(argv[1], "rb") : nullptr;
    if (!f || fread(hu.data(), 4, n3, f) != n3) { printf("need a displacement dump of %d x 3 x 128^3 floats\n", B); return 1; }
    fclose(f);
    for (size_t i = 0; i < n3; ++i) hv[i] = sinf(0.001f * (float)(i % 100003));
    float *out, *ref, *img, *u;
    (void)hipMalloc(&out, n3 * 4); (void)hipMalloc(&ref, n3 * 4); (void)hipMalloc(&img, n3 * 4); (void)hipMalloc(&u, n3 * 4);
    (void)hipMemcpy(u, hu.data(), n3 * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(img, hv.data(), n3 * 4, hipMemcpyHostToDevice);
    if (argc > 3) {  // argv[3]: "identity" = zero displacement, "small" = scale the displacement by 0.25
        const float sc = argv[3][0] == 'i' ? 0.f : 0.25f;
        for (size_t i = 0; i < n3; ++i) hu[i] *= sc;
        (void)hipMemcpy(u, hu.data(), n3 * 4, hipMemcpyHostToDevice);
        printf("displacement scaled by %.2f\n", sc);
    }
    const double bytes = 36.0 * B * nv;
    const float t0 = run<0, 2>(ref, img, u, B);
    std::vector<float> hr(n3), ho(n3);
    (void)hipMemcpy(hr.data(), ref, n3 * 4, hipMemcpyDeviceToHost);
    auto check = [&](const char* name, float us) {
        (void)hipMemcpy(ho.data(), out, n3 * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < n3; ++i) bad += ho[i] != hr[i];
        printf("%-34s %8.1f us  %6.0f GB/s  %s\n", name, us, bytes / us / 1e3, bad ? "MISMATCH" : "bits ok");
    };
    printf("interp_forward C=3, batch %d x 128^3 (36 B/voxel)\n", B);
    printf("%-34s %8.1f us  %6.0f GB/s\n", "V0 pair loads, U=2", t0, bytes / t0 / 1e3);
    check("V0 pair loads, U=1", run<0, 1>(out, img, u, B));
    check("V0 pair loads, U=4", run<0, 4>(out, img, u, B));
    check("V1 interior fast path, U=2", run<1, 2>(out, img, u, B));
    check("V2 eight dword loads, U=2", run<2, 2>(out, img, u, B));
    check("V3 dword + DPP share, U=2", run<3, 2>(out, img, u, B));
    check("V3 dword + DPP share, U=1", run<3, 1>(out, img, u, B));
    check("V3 dword + DPP share, U=4", run<3, 4>(out, img, u, B));
    check("V4 12 pair loads in flight, U=2", run<4, 2>(out, img, u, B));
    check("V4 12 pair loads in flight, U=1", run<4, 1>(out, img, u, B));
    { const float a6 = run<6, 2>(out, img, u, B), a7 = run<7, 2>(out, img, u, B);
      printf("V6 no image access %.1f us (%.0f GB/s of 24 B/voxel)   V7 coalesced image read %.1f us (%.0f GB/s of 36 B/voxel)\n", a6, 24.0 * B * nv / a6 / 1e3, a7, bytes / a7 / 1e3); }
    { const float a9 = run<9, 2>(out, img, u, B), a10 = run<10, 2>(out, img, u, B);
      printf("V9 gathers, trivial arithmetic %.1f us   V10 coalesced read, full select + trilinear arithmetic %.1f us\n", a9, a10); }
    check("V8 LDS 4x8 rows M=1, 1024 thr", runl<4, 8, 1, 1024>(out, img, u, B));
    check("V8 LDS 4x8 rows M=2, 1024 thr", runl<4, 8, 2, 1024>(out, img, u, B));
    check("V8 LDS 4x4 rows M=1, 512 thr", runl<4, 4, 1, 512>(out, img, u, B));
    check("V8 LDS 4x4 rows M=1, 1024 thr", runl<4, 4, 1, 1024>(out, img, u, B));
    check("V8 LDS 8x8 rows M=1, 1024 thr", runl<8, 8, 1, 1024>(out, img, u, B));
    check("V8 LDS 4x8 rows M=1, 512 thr", runl<4, 8, 1, 512>(out, img, u, B));
    check("V8 LDS 2x8 rows M=1, 1024 thr", runl<2, 8, 1, 1024>(out, img, u, B));
    check("V5 tile 1x4 rows, 256 thr", runt<1, 4, 256>(out, img, u, B));
    check("V5 tile 2x2 rows, 256 thr", runt<2, 2, 256>(out, img, u, B));
    check("V5 tile 2x4 rows, 256 thr", runt<2, 4, 256>(out, img, u, B));
    check("V5 tile 4x4 rows, 256 thr", runt<4, 4, 256>(out, img, u, B));
    check("V5 tile 2x4 rows, 512 thr", runt<2, 4, 512>(out, img, u, B));
    check("V5 tile 4x4 rows, 512 thr", runt<4, 4, 512>(out, img, u, B));
    check("V5 tile 4x4 rows, 1024 thr", runt<4, 4, 1024>(out, img, u, B));
    check("V5 tile 4x8 rows, 1024 thr", runt<4, 8, 1024>(out, img, u, B));
    check("V5 tile 8x8 rows, 1024 thr", runt<8, 8, 1024>(out, img, u, B));
    check("V5 tile 2x8 rows, 1024 thr", runt<2, 8, 1024>(out, img, u, B));
    return 0;
}
