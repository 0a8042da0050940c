// Shared device helpers and launch plumbing for liblagomorph_hip (gfx950 only).
//
// Layout everywhere: N C (D) H W, last spatial axis fastest.  Kernels assign one
// lane to one voxel of the flattened spatial index s = (i*ny + j)*nz + k of one
// batch item, so a 64-lane wavefront touches 256 contiguous bytes of every
// streamed operand.  Workgroups are re-ordered so that each XCD walks one
// contiguous eighth of the launch (its own L2 then holds the +-1 row / +-1 slab
// neighbours that stencils and gathers re-read).
//
// Arithmetic follows the reference expression by expression (cited per helper);
// the library is built with -ffp-contract=off so that results are bit-identical
// to the strict-IEEE CPU oracle wherever no atomic is involved.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/lagomorph_hip.h"

namespace lago {

constexpr int kBlock = 256;

// ---------------------------------------------------------------- host side

int fail_invalid(const char *fmt, ...);
int fail_hip(hipError_t e, const char *what);
int finish_launch(hipStream_t s, const char *what);  // hipGetLastError (+ sync in debug mode)
extern int g_splat_mode;

#define LAGO_HIP_TRY(expr)                                      \
    do {                                                        \
        hipError_t e__ = (expr);                                \
        if (e__ != hipSuccess) return lago::fail_hip(e__, #expr); \
    } while (0)

// Unsigned division by a runtime-constant divisor (n < 2^31): q = (mulhi(n, m) + n) >> l.
struct FastDiv {
    uint32_t d, m, l;
    FastDiv() : d(1), m(1), l(0) {}
    explicit FastDiv(uint32_t dd) : d(dd) {
        l = 0;
        while ((1ull << l) < dd) ++l;
        m = (uint32_t)(((1ull << 32) * ((1ull << l) - dd)) / dd + 1);
    }
    __host__ __device__ __forceinline__ uint32_t div(uint32_t n) const {
#if defined(__HIP_DEVICE_COMPILE__)
        return (__umulhi(n, m) + n) >> l;
#else
        return (uint32_t)((((uint64_t)n * m) >> 32) + n) >> l;
#endif
    }
};

// Spatial geometry of one batch item plus the launch decomposition.
struct Geom {
    int nx, ny, nz;     // 2D fields use nx = 1, (ny, nz) = (H, W): same flattened layout
    uint32_t nvox;      // nx*ny*nz
    uint32_t nbx;       // workgroups per batch item
    uint32_t nblocks;   // nbx * nn
    FastDiv dyz, dz, dnbx;
};

inline bool make_geom(Geom &g, int dim, int64_t nn, int64_t nx, int64_t ny, int64_t nz, int vox_per_block = kBlock) {
    if (dim == 2) {
        nz = ny;
        ny = nx;
        nx = 1;
    }
    if (nn < 0 || nx < 1 || ny < 1 || nz < 1) return false;
    int64_t nv = nx * ny * nz;
    if (nv >= (1ll << 31)) return false;
    g.nx = (int)nx;
    g.ny = (int)ny;
    g.nz = (int)nz;
    g.nvox = (uint32_t)nv;
    g.nbx = (uint32_t)((nv + vox_per_block - 1) / vox_per_block);
    int64_t nb = (int64_t)g.nbx * nn;
    if (nb >= (1ll << 31)) return false;
    g.nblocks = (uint32_t)nb;
    g.dyz = FastDiv((uint32_t)(ny * nz));
    g.dz = FastDiv((uint32_t)nz);
    g.dnbx = FastDiv(g.nbx);
    return true;
}

// ---------------------------------------------------------------- device side

// Workgroups are dealt round-robin over the 8 XCDs (block b -> XCD b % 8).  Map
// physical block b to logical block L so that XCD x owns the contiguous range
// [x*q, (x+1)*q): neighbours in memory then share one L2.  Speed only.
__device__ __forceinline__ uint32_t xcd_swizzle(uint32_t b, uint32_t total) {
    uint32_t q = total >> 3;
    if (b >= (q << 3)) return b;
    return (b & 7u) * q + (b >> 3);
}

struct Vox {
    uint32_t n;  // batch item
    uint32_t s;  // flattened spatial index
    int i, j, k;
    bool valid;
};

__device__ __forceinline__ Vox locate(const Geom &g) {
    Vox v;
    uint32_t L = xcd_swizzle(blockIdx.x, g.nblocks);
    v.n = g.dnbx.div(L);
    uint32_t bx = L - v.n * g.nbx;
    v.s = bx * kBlock + threadIdx.x;
    v.valid = v.s < g.nvox;
    uint32_t s = v.valid ? v.s : 0;
    uint32_t i = g.dyz.div(s);
    uint32_t r = s - i * (uint32_t)(g.ny * g.nz);
    uint32_t j = g.dz.div(r);
    v.i = (int)i;
    v.j = (int)j;
    v.k = (int)(r - j * (uint32_t)g.nz);
    return v;
}

// include/interp.h:64-70 (floor rule), saturated like the oracle.
template <typename R>
__device__ __forceinline__ int lg_floor(R x) {
    x = x > (R)1073741824.0 ? (R)1073741824.0 : x;
    x = x < (R)-1073741824.0 ? (R)-1073741824.0 : x;
    int f = (int)x;
    if (x < 0 && x != (R)f) --f;
    return f;
}

// include/extrap.h:46-57 clampBackground on a (floor, ceil) pair
__device__ __forceinline__ void clamp_pair(int &fl, int &ce, int size) {
    if (fl < 0) {
        fl = 0;
        if (ce < 0) ce = 0;
    }
    if (ce >= size) {
        ce = size - 1;
        if (fl >= size) fl = size - 1;
    }
}

// include/extrap.h:41-44
__device__ __forceinline__ int clamp1(int r, int b) { return r < 0 ? 0 : (r >= b ? b - 1 : r); }

// Sample position x + dt*u: computed in double (dt is a double in the
// reference, cuda/interp.cu:36-37,68-70) and narrowed to R.
template <typename R>
__device__ __forceinline__ R sample_pos(int i, double dt, R u) {
    return (R)((double)(R)i + dt * (double)u);
}

// 2D/3D lerp stencils: clamped corner offsets plus the fractional parts.
template <typename R>
struct Lerp3 {
    uint32_t o[8];  // v0..v7 in the reference's corner order (include/interp.h:91-98)
    R t, u, v;
    __device__ __forceinline__ void setup(R x, R y, R z, int sx, int sy, int sz) {
        int fx = lg_floor(x), fy = lg_floor(y), fz = lg_floor(z);
        int cx = fx + 1, cy = fy + 1, cz = fz + 1;
        t = x - (R)fx;
        u = y - (R)fy;
        v = z - (R)fz;
        clamp_pair(fx, cx, sx);
        clamp_pair(fy, cy, sy);
        clamp_pair(fz, cz, sz);
        uint32_t ff = ((uint32_t)fx * sy + fy) * sz, cf = ((uint32_t)cx * sy + fy) * sz;
        uint32_t cc = ((uint32_t)cx * sy + cy) * sz, fc = ((uint32_t)fx * sy + cy) * sz;
        o[0] = ff + fz; o[1] = cf + fz; o[2] = cc + fz; o[3] = fc + fz;
        o[4] = ff + cz; o[5] = cf + cz; o[6] = cc + cz; o[7] = fc + cz;
    }
    // include/interp.h:115-122
    __device__ __forceinline__ R value(const R *__restrict__ img) const {
        R omt = (R)1.f - t, omu = (R)1.f - u, omv = (R)1.f - v;
        R v0 = img[o[0]], v1 = img[o[1]], v2 = img[o[2]], v3 = img[o[3]];
        R v4 = img[o[4]], v5 = img[o[5]], v6 = img[o[6]], v7 = img[o[7]];
        return omv * (omu * (omt * v0 + t * v1) + u * (omt * v3 + t * v2)) +
               v * (omu * (omt * v4 + t * v5) + u * (omt * v7 + t * v6));
    }
    // include/interp.h:315-326
    __device__ __forceinline__ void grad(const R *__restrict__ img, R &gx, R &gy, R &gz) const {
        R omt = (R)1.f - t, omu = (R)1.f - u, omv = (R)1.f - v;
        R v0 = img[o[0]], v1 = img[o[1]], v2 = img[o[2]], v3 = img[o[3]];
        R v4 = img[o[4]], v5 = img[o[5]], v6 = img[o[6]], v7 = img[o[7]];
        gx = omv * (omu * (v1 - v0) + u * (v2 - v3)) + v * (omu * (v5 - v4) + u * (v6 - v7));
        gy = omv * (omt * (v3 - v0) + t * (v2 - v1)) + v * (omt * (v7 - v4) + t * (v6 - v5));
        gz = omu * (omt * (v4 - v0) + t * (v5 - v1)) + u * (omt * (v7 - v3) + t * (v6 - v2));
    }
};

template <typename R>
struct Lerp2 {
    uint32_t o[4];  // v0..v3 (include/interp.h:36-39)
    R t, u;
    __device__ __forceinline__ void setup(R x, R y, int sx, int sy) {
        int fx = lg_floor(x), fy = lg_floor(y);
        int cx = fx + 1, cy = fy + 1;
        t = x - (R)fx;
        u = y - (R)fy;
        clamp_pair(fx, cx, sx);
        clamp_pair(fy, cy, sy);
        o[0] = (uint32_t)fx * sy + fy;
        o[1] = (uint32_t)cx * sy + fy;
        o[2] = (uint32_t)cx * sy + cy;
        o[3] = (uint32_t)fx * sy + cy;
    }
    // include/interp.h:52-55
    __device__ __forceinline__ R value(const R *__restrict__ img) const {
        R omt = (R)1.f - t, omu = (R)1.f - u;
        R v0 = img[o[0]], v1 = img[o[1]], v2 = img[o[2]], v3 = img[o[3]];
        return omt * (omu * v0 + u * v3) + t * (omu * v1 + u * v2);
    }
    // include/interp.h:202-203
    __device__ __forceinline__ void grad(const R *__restrict__ img, R &gx, R &gy) const {
        R v0 = img[o[0]], v1 = img[o[1]], v2 = img[o[2]], v3 = img[o[3]];
        gx = v1 - v0 + u * (v2 - v3 - v1 + v0);
        gy = v3 - v0 + t * (v2 - v1 - v3 + v0);
    }
};

// Splat stencils: the reference's sequentially flipped weights
// (include/interp.h:404-454) and per-corner clamped indices (:376-380).
template <typename R>
struct Splat3 {
    uint32_t o[8];  // corner order of the reference loop: x outer, y, z inner
    R w[8];
    int x0, y0, z0;  // unclamped floor corner
    __device__ __forceinline__ void setup(R x, R y, R z, int sx, int sy, int sz) {
        x0 = lg_floor(x);
        y0 = lg_floor(y);
        z0 = lg_floor(z);
        R dx = (R)1.f - (x - (R)x0);
        R dy = (R)1.f - (y - (R)y0);
        R dz = (R)1.f - (z - (R)z0);
        int q = 0;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    int i = clamp1(x0 + a, sx), j = clamp1(y0 + b, sy), k = clamp1(z0 + c, sz);
                    w[q] = dx * dy * dz;
                    o[q] = ((uint32_t)i * sy + j) * sz + k;
                    ++q;
                    dz = (R)1.f - dz;
                }
                dy = (R)1.f - dy;
            }
            dx = (R)1.f - dx;
        }
    }
};

template <typename R>
struct Splat2 {
    uint32_t o[4];
    R w[4];
    __device__ __forceinline__ void setup(R x, R y, int sx, int sy) {
        int x0 = lg_floor(x), y0 = lg_floor(y);
        R dx = (R)1.f - (x - (R)x0);
        R dy = (R)1.f - (y - (R)y0);
        int q = 0;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                int i = clamp1(x0 + a, sx), j = clamp1(y0 + b, sy);
                w[q] = dx * dy;
                o[q] = (uint32_t)i * sy + j;
                ++q;
                dy = (R)1.f - dy;
            }
            dx = (R)1.f - dx;
        }
    }
};

// No-return hardware float atomics (global_atomic_add_f32 / _f64 on gfx950).
__device__ __forceinline__ void atomic_add(float *p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add(double *p, double v) { unsafeAtomicAdd(p, v); }

}  // namespace lago
