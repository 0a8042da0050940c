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
#include <atomic>
#include <array>
#include <mutex>
#include <stdint.h>
#include <stdio.h>

#include "../../include/lagomorph_hip.h"

namespace lago {

constexpr int kBlock = 256;

// ---------------------------------------------------------------- host side

int fail_invalid(const char *fmt, ...);
int fail_hip(hipError_t e, const char *what);
int finish_launch(hipStream_t s, const char *what);  // hipGetLastError (+ sync in debug mode)
// process-wide tuning settings (speed only, never results): atomics, because autograd runs the backward entry
// points on threads of its own while a host thread may change a setting
template <int N>
struct KnobArray {  // array-valued setting: read and written as a whole under a mutex
    std::mutex mu;
    std::array<int, N> v;
    KnobArray(std::array<int, N> init) : v(init) {}
    std::array<int, N> get() {
        std::lock_guard<std::mutex> l(mu);
        return v;
    }
    void set(std::array<int, N> nv) {
        std::lock_guard<std::mutex> l(mu);
        v = nv;
    }
};
// Launch direction.  The 256 MB Infinity Cache (memory side) still holds the END of what the previous kernel wrote:
// a consumer that walks its blocks in the REVERSE of its producer's order starts on cached data (streaming chain of
// 805 MB buffers: 5.54 -> 5.90 TB/s, 256 MB: 5.38 -> 6.03; tools/probes/mall_order.hip).  Every launch of the big
// kernels therefore takes the opposite direction of the launch before it (0 = ascending logical blocks).  Speed
// only: the block -> data mapping is a bijection either way.
extern std::atomic<int> g_launch_alt;        // 1 (default): alternate; 0: always ascending
extern std::atomic<unsigned> g_launch_seq;
extern std::atomic<long long> g_reversed_launches;   // telemetry: lago_reversed_launches()
inline int next_direction() {
    const int rev = g_launch_alt ? (int)(g_launch_seq.fetch_add(1) & 1u) : 0;
    if (rev) g_reversed_launches.fetch_add(1, std::memory_order_relaxed);
    return rev;
}
extern std::atomic<int> g_splat_mode;
extern std::atomic<int> g_interp_vec;  // 1: use the vectorised 3D kernels when shapes allow (default)
// Which implementation a call was dispatched to (lago_path_launches; ids = LAGO_PATH_* of the header).  Telemetry only:
// the tests use it to make sure a case meant to exercise a fast path really runs it.
enum { LP_GATHER_WINDOW = 0, LP_STENCIL_TILE, LP_VECTOR_GATHER, LP_SPLAT_SHEAR, LP_SPLAT_SHEAR_MC, LP_SPLAT_TILED,
       LP_SPLAT_GLOBAL, LP_FLUID_LDS, LP_FLUID_2D, LP_FLUID_XPASS, LP_FLUID_ROCFFT, LP_SPLAT_2D, LP_SPLAT_AFFINE_BOX, LP_FLUID_GENERIC, LP_COUNT };
extern std::atomic<long long> g_path_launches[LP_COUNT];
inline void note_path(int p) { g_path_launches[p].fetch_add(1, std::memory_order_relaxed); }

// tuning (include/lagomorph_hip.h: lago_tuning): every module applies its own fields (api.hip: lago_set_tuning)
void tune_splat(const int32_t *tile7, const int32_t *shear8, int shear_mc, int mc);    // splat.hip
void tune_fused(int stencil_tile, int gather_window);                                   // fused.hip
void tune_fluid(int mode);                                                              // fft.hip
void tune_fluid_passes(int ipw, int zy_persist, int xpass_wide, int xpass_persist);     // fft3.hip
void tune_affine(int box);                                                              // affine.hip

#define LAGO_HIP_TRY(expr)                                      \
    do {                                                        \
        hipError_t e__ = (expr);                                \
        if (e__ != hipSuccess) return lago::fail_hip(e__, #expr); \
    } while (0)

// Unsigned division by a runtime-constant divisor (n < 2^31): q = (mulhi(n, m) + n) >> l.
struct FastDiv {
    uint32_t d, m, l;
    __host__ __device__ constexpr FastDiv() : d(1), m(1), l(0) {}
    __host__ __device__ constexpr explicit FastDiv(uint32_t dd) : d(dd), m(0), l(0) {
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
    int rev;            // launch direction (next_direction()): logical blocks descending
    FastDiv dyz, dz, dnbx;
};

// scatter = true: the launch feeds a scatter-add (splats, the d_A / d_T reductions).  Float atomics round in arrival
// order, so the block order of such a launch shows in the last bits of its result: it is then always ascending and the
// call leaves the alternation counter alone -- the bits of a scatter-add never depend on which calls (of any kind, valid
// or rejected) this process has made before (tests/test_gpu_dispatch.py: test_scatter_direction_is_history_free).
inline bool make_geom(Geom &g, int dim, int64_t nn, int64_t nx, int64_t ny, int64_t nz, int vox_per_block = kBlock,
                      bool scatter = false) {
    // one channel of one batch item is addressed with 32-bit byte offsets: nvox * 8 < 2^32
    if (dim == 2) {
        nz = ny;
        ny = nx;
        nx = 1;
    }
    if (nn < 0 || nx < 1 || ny < 1 || nz < 1) return false;
    int64_t nv = nx * ny * nz;
    if (nv >= (1ll << 29)) return false;
    g.nx = (int)nx;
    g.ny = (int)ny;
    g.nz = (int)nz;
    g.nvox = (uint32_t)nv;
    g.nbx = (uint32_t)((nv + vox_per_block - 1) / vox_per_block);
    int64_t nb = (int64_t)g.nbx * nn;
    if (nb >= (1ll << 31)) return false;
    g.nblocks = (uint32_t)nb;
    g.rev = scatter ? 0 : next_direction();
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
// ... and in the launch's direction (Geom::rev / next_direction())
__device__ __forceinline__ uint32_t block_order(uint32_t b, uint32_t total, int rev) {
    const uint32_t L = xcd_swizzle(b, total);
    return rev ? total - 1u - L : L;
}

struct Vox {
    uint32_t n;  // batch item
    uint32_t s;  // flattened spatial index
    int i, j, k;
    bool valid;
};

__device__ __forceinline__ Vox locate(const Geom &g) {
    Vox v;
    uint32_t L = block_order(blockIdx.x, g.nblocks, g.rev);
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

// ---- arithmetic contract ----------------------------------------------------
// Every `a*b + c` of the reference is evaluated as one fused multiply-add and a
// sum of two products `a*b + c*d` as fma(a, b, c*d) (left product fused): the
// contraction nvcc applies by default to the reference.  The CPU oracle spells
// out the identical pattern (oracle/lago_oracle_impl.h, LG_FMA), and the library
// is compiled with -ffp-contract=off so that nothing else fuses.
__device__ __forceinline__ float lg_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double lg_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// include/interp.h:64-70: (int)x, minus one for negative non-integers == floor(x).
// Saturated to +-2^30 (as the oracle does) so that floor + 1 cannot overflow.
// float: one v_cvt_flr_i32_f32 (floor + convert, saturating at the int32 range) and one
// v_med3_i32 -- the same value as saturating first, for every finite x.
__device__ __forceinline__ int lg_med3(int x, int lo, int hi) {  // lo <= hi
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ int lg_floor(float x) {
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return lg_med3(r, -1073741824, 1073741824);
}
__device__ __forceinline__ int lg_floor(double x) {
    x = x > 1073741824.0 ? 1073741824.0 : x;
    x = x < -1073741824.0 ? -1073741824.0 : x;
    return (int)__builtin_floor(x);
}

// include/extrap.h:41-44 clamp(); clampBackground (extrap.h:46-57) on a
// (floor, floor+1) pair is this clamp applied to both members.
__device__ __forceinline__ int clamp1(int r, int b) { return lg_med3(r, 0, b - 1); }  // b >= 1

// Sample position x + dt*u: computed in double (dt is a double in the
// reference, cuda/interp.cu:36-37,68-70) and narrowed to R.
// Cost note: the double path is 4 half-rate instructions per coordinate.  For dt == +-1 (the
// default of deform.interp) the double result is the exactly rounded sum i +- u, which a single
// float add produces bit for bit (i < 2^24 is exact in float, one rounding either way), so that
// case skips the conversions.  `dt` is wave-uniform: the branch is scalar.
template <typename R>
__device__ __forceinline__ R sample_pos(int i, double dt, R u) {
    if (sizeof(R) == 4) {
        if (dt == 1.0) return (R)((float)i + (float)u);
        if (dt == -1.0) return (R)((float)i - (float)u);
    }
    return (R)__builtin_fma(dt, (double)u, (double)i);
}

// Unrolled kernels hoist the dt == +-1 decision to a template parameter (float only): the
// position is fma(+-1, u, i), one rounding of the exact sum, as above.
template <typename R, bool UNIT>
__device__ __forceinline__ R sample_pos_t(int i, double dt, R u) {
    if (UNIT && sizeof(R) == 4) return (R)__builtin_fmaf((float)dt, (float)u, (float)i);
    return (R)__builtin_fma(dt, (double)u, (double)i);
}
template <typename R>
__host__ __device__ inline bool unit_dt(double dt) { return sizeof(R) == 4 && (dt == 1.0 || dt == -1.0); }

// Gathers go through buffer loads: a 128-bit descriptor in SGPRs (built from a
// wave-uniform plane pointer) plus a 32-bit per-lane byte offset -- no 64-bit
// address arithmetic per load, and out-of-range offsets read 0 instead of faulting.
typedef __amdgpu_buffer_rsrc_t BufRsrc;
typedef unsigned int lg_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int lg_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ BufRsrc make_rsrc(const void *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
// A store with a cache policy chosen at build time: NT = 1 marks it non-temporal (streamed through the L2, evicted
// first), which keeps a kernel's own output from displacing the lines its gathers / stencil neighbours re-read
// (profiles/r04_cache_policy.md).
template <int NT, typename R>
__device__ __forceinline__ R ld_pol(const R *p) {   // (the same for a load of data that is read once)
    if constexpr (NT != 0) return __builtin_nontemporal_load(p);
    else return *p;
}
template <int NT, typename R>
__device__ __forceinline__ void st_pol(R *p, R v) {
    if constexpr (NT != 0) __builtin_nontemporal_store(v, p);
    else *p = v;
}

template <typename R>
__device__ __forceinline__ R buf_load1(BufRsrc r, uint32_t off);
template <>
__device__ __forceinline__ float buf_load1<float>(BufRsrc r, uint32_t off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
template <>
__device__ __forceinline__ double buf_load1<double>(BufRsrc r, uint32_t off) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
}
// two adjacent elements with ONE load (element-aligned, not pair-aligned)
template <typename R>
__device__ __forceinline__ void buf_load2(BufRsrc r, uint32_t off, R &lo, R &hi);
// hipcc 7.2 narrows a b64/b128 buffer load to its first dword when the vector elements are
// extracted and only consumed through selects (observed miscompile; tools/probes/bufload_align.hip).
// Going through a scalar integer of the full width avoids the pattern.
template <>
__device__ __forceinline__ void buf_load2<float>(BufRsrc r, uint32_t off, float &lo, float &hi) {
    const lg_u32x2 p = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
    const unsigned long long q = __builtin_bit_cast(unsigned long long, p);
    lo = __builtin_bit_cast(float, (unsigned int)q);
    hi = __builtin_bit_cast(float, (unsigned int)(q >> 32));
}
// the same with an extra wave-uniform byte offset in the instruction's scalar-offset field
__device__ __forceinline__ void buf_load2s(BufRsrc r, uint32_t off, uint32_t soff, float &lo, float &hi) {
    const lg_u32x2 p = __builtin_amdgcn_raw_buffer_load_b64(r, off, soff, 0);
    const unsigned long long q = __builtin_bit_cast(unsigned long long, p);
    lo = __builtin_bit_cast(float, (unsigned int)q);
    hi = __builtin_bit_cast(float, (unsigned int)(q >> 32));
}
struct lg_u64x2 {
    unsigned long long a, b;
};
template <>
__device__ __forceinline__ void buf_load2<double>(BufRsrc r, uint32_t off, double &lo, double &hi) {
    const lg_u32x4 p = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    const lg_u64x2 q = __builtin_bit_cast(lg_u64x2, p);
    lo = __builtin_bit_cast(double, q.a);
    hi = __builtin_bit_cast(double, q.b);
}

// Trilinear stencil.  The 8 corners are 4 (x, y) rows times the z pair
// (floor, floor+1), which is contiguous in memory: each row is fetched with ONE
// pair load at z = zb = min(clamped floor, nz-2); when the sample lies beyond a z
// border both members of the clamped pair coincide and are picked from the same
// half.  Offsets are 32-bit byte offsets from a wave-uniform base pointer, so the
// loads use the scalar-base addressing mode and need no 64-bit address math.
// THIN_OK = false promises nz >= 2 at compile time (the vectorised kernels): without the
// per-sample `thin` branch the compiler batches the pair loads of several samples.
template <typename R, bool THIN_OK = true>
struct Lerp3 {
    uint32_t rb[4];   // byte offsets of rows (fx,fy) (cx,fy) (cx,cy) (fx,cy) at z = zb: v0/v4 v1/v5 v2/v6 v3/v7
    R t, u, v;
    bool f_hi, c_lo;  // floor-z value is the pair's .hi / ceil-z value is the pair's .lo
    bool thin;        // nz == 1: no pair exists
    uint32_t bytes;   // size of one channel plane (uniform)
    __device__ __forceinline__ void setup(R x, R y, R z, int sx, int sy, int sz) {
        const int flx = lg_floor(x), fly = lg_floor(y), flz = lg_floor(z);
        t = x - (R)flx;
        u = y - (R)fly;
        v = z - (R)flz;
        const int fx = clamp1(flx, sx), cx = clamp1(flx + 1, sx);
        const int fy = clamp1(fly, sy), cy = clamp1(fly + 1, sy);
        const int fz = clamp1(flz, sz), cz = clamp1(flz + 1, sz);
        thin = THIN_OK && sz < 2;
        const int zb = thin ? 0 : min(fz, sz - 2);
        f_hi = fz != zb;
        c_lo = cz == zb;
        const uint32_t rowB = (uint32_t)sz * (uint32_t)sizeof(R);   // uniform
        const uint32_t slabB = (uint32_t)sy * rowB;                  // uniform
        bytes = (uint32_t)sx * slabB;
        // 24-bit multiplies are full rate, 32-bit ones quarter rate; slabB < 2^24 holds for every
        // volume whose (y, z) plane is below 16 MiB, the branch is wave-uniform
        const uint32_t ff = slabB < (1u << 24)
                                ? __umul24((uint32_t)fx, slabB) + __umul24((uint32_t)fy, rowB) + (uint32_t)zb * (uint32_t)sizeof(R)
                                : ((uint32_t)fx * (uint32_t)sy + (uint32_t)fy) * rowB + (uint32_t)zb * (uint32_t)sizeof(R);
        const uint32_t dX = cx != fx ? slabB : 0u;
        const uint32_t dY = cy != fy ? rowB : 0u;
        rb[0] = ff;
        rb[1] = ff + dX;
        rb[2] = ff + dX + dY;
        rb[3] = ff + dY;
    }
    // `img` must be wave-uniform (one channel plane): it becomes a buffer
    // descriptor held in SGPRs, the per-lane part is the 32-bit byte offset.
    __device__ __forceinline__ void fetch(const R *__restrict__ img, R (&c)[8]) const {
        const BufRsrc r = make_rsrc(img, bytes);
        if (THIN_OK && thin) {
#pragma unroll
            for (int q = 0; q < 4; ++q) c[q] = c[q + 4] = buf_load1<R>(r, rb[q]);
            return;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            R lo, hi;
            buf_load2<R>(r, rb[q], lo, hi);
            c[q] = f_hi ? hi : lo;
            c[q + 4] = c_lo ? lo : hi;
        }
    }
    // include/interp.h:115-122.  The z-border selection is made AFTER the two (x, y) interpolations: the floor-z group
    // c0..c3 is either the four `lo` halves or the four `hi` halves (f_hi is one flag for all four rows), the ceil-z group
    // likewise, and both groups go through the same expression G -- so fma(omv, f_hi ? G(hi) : G(lo), v * (c_lo ? G(lo) :
    // G(hi))) is value_of(c) bit for bit with two selects instead of eight (the gather kernels are half bound by vector
    // instruction issue: 48 of Ad_star's 424 vector instructions per thread were these selects).
    __device__ __forceinline__ R value(const R *__restrict__ img) const {
        if (THIN_OK && thin) {
            R c[8];
            fetch(img, c);
            return value_of(c);
        }
        const BufRsrc r = make_rsrc(img, bytes);
        R lo[4], hi[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) buf_load2<R>(r, rb[q], lo[q], hi[q]);
        const R omt = (R)1.f - t, omu = (R)1.f - u, omv = (R)1.f - v;
        const R glo = lg_fma(omu, lg_fma(omt, lo[0], t * lo[1]), u * lg_fma(omt, lo[3], t * lo[2]));
        const R ghi = lg_fma(omu, lg_fma(omt, hi[0], t * hi[1]), u * lg_fma(omt, hi[3], t * hi[2]));
        return lg_fma(omv, f_hi ? ghi : glo, v * (c_lo ? glo : ghi));
    }
    __device__ __forceinline__ R value_of(const R (&c)[8]) const {
        const R omt = (R)1.f - t, omu = (R)1.f - u, omv = (R)1.f - v;
        return lg_fma(omv, lg_fma(omu, lg_fma(omt, c[0], t * c[1]), u * lg_fma(omt, c[3], t * c[2])),
                      v * lg_fma(omu, lg_fma(omt, c[4], t * c[5]), u * lg_fma(omt, c[7], t * c[6])));
    }
    // include/interp.h:315-326
    __device__ __forceinline__ void grad(const R *__restrict__ img, R &gx, R &gy, R &gz) const {
        R c[8];
        fetch(img, c);
        const R omt = (R)1.f - t, omu = (R)1.f - u, omv = (R)1.f - v;
        gx = lg_fma(omv, lg_fma(omu, c[1] - c[0], u * (c[2] - c[3])), v * lg_fma(omu, c[5] - c[4], u * (c[6] - c[7])));
        gy = lg_fma(omv, lg_fma(omt, c[3] - c[0], t * (c[2] - c[1])), v * lg_fma(omt, c[7] - c[4], t * (c[6] - c[5])));
        gz = lg_fma(omu, lg_fma(omt, c[4] - c[0], t * (c[5] - c[1])), u * lg_fma(omt, c[7] - c[3], t * (c[6] - c[2])));
    }
};

template <typename R>
struct Lerp2 {
    uint32_t o[4];  // v0..v3 (include/interp.h:36-39)
    R t, u;
    __device__ __forceinline__ void setup(R x, R y, int sx, int sy) {
        const int flx = lg_floor(x), fly = lg_floor(y);
        t = x - (R)flx;
        u = y - (R)fly;
        const int fx = clamp1(flx, sx), cx = clamp1(flx + 1, sx);
        const int fy = clamp1(fly, sy), cy = clamp1(fly + 1, sy);
        o[0] = (uint32_t)fx * sy + fy;
        o[1] = (uint32_t)cx * sy + fy;
        o[2] = (uint32_t)cx * sy + cy;
        o[3] = (uint32_t)fx * sy + cy;
    }
    // include/interp.h:52-55
    __device__ __forceinline__ R value(const R *__restrict__ img) const {
        const R omt = (R)1.f - t, omu = (R)1.f - u;
        const R v0 = img[o[0]], v1 = img[o[1]], v2 = img[o[2]], v3 = img[o[3]];
        return lg_fma(omt, lg_fma(omu, v0, u * v3), t * lg_fma(omu, v1, u * v2));
    }
    // include/interp.h:202-203
    __device__ __forceinline__ void grad(const R *__restrict__ img, R &gx, R &gy) const {
        const R v0 = img[o[0]], v1 = img[o[1]], v2 = img[o[2]], v3 = img[o[3]];
        gx = lg_fma(u, v2 - v3 - v1 + v0, v1 - v0);
        gy = lg_fma(t, v2 - v1 - v3 + v0, v3 - v0);
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

// affine_interp_backward's image splat has two kernels (affine.hip: affine_splat_box_kernel for "regular" matrices --
// invertible, with an inverse that does not blow a target box up beyond 4 x per axis -- and splat.hip's general tiled
// kernel for the others); both are always launched and every workgroup decides by THIS function, from the matrix in
// device memory, whether the batch item is its own (no host synchronisation).  Ai: the inverse (valid when true).
template <typename R>
__device__ __forceinline__ bool affine_item_regular(const R *An, double (&Ai)[9]) {
    double a[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) a[q] = (double)An[q];
    const double c0 = a[4] * a[8] - a[5] * a[7], c1 = a[5] * a[6] - a[3] * a[8], c2 = a[3] * a[7] - a[4] * a[6];
    const double det = a[0] * c0 + a[1] * c1 + a[2] * c2;
    double big = 0;
#pragma unroll
    for (int q = 0; q < 9; ++q) big = fabs(a[q]) > big ? fabs(a[q]) : big;
    if (!(fabs(det) > 1e-3) || !(big < 1e3)) return false;   // (NaN fails both)
    const double id = 1.0 / det;
    Ai[0] = c0 * id; Ai[1] = (a[2] * a[7] - a[1] * a[8]) * id; Ai[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    Ai[3] = c1 * id; Ai[4] = (a[0] * a[8] - a[2] * a[6]) * id; Ai[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    Ai[6] = c2 * id; Ai[7] = (a[1] * a[6] - a[0] * a[7]) * id; Ai[8] = (a[0] * a[4] - a[1] * a[3]) * id;
    double rows = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double rs = fabs(Ai[3 * r]) + fabs(Ai[3 * r + 1]) + fabs(Ai[3 * r + 2]);
        rows = rs > rows ? rs : rows;
    }
    return rows <= 4.0;
}

// No-return hardware float atomics (global_atomic_add_f32 / _f64 on gfx950).
__device__ __forceinline__ void atomic_add(float *p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add(double *p, double v) { unsafeAtomicAdd(p, v); }

}  // namespace lago
