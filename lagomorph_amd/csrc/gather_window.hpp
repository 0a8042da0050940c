// Trilinear gathers through an LDS window (float, 3D).
//
// The pair-gather kernels (Lerp3::fetch) fetch the eight corners of every sample with four dwordx2 loads
// through the vector L1; the TCP returns 32 B per clock for those whatever the address pattern
// (tools/probes/l1_gather_rate.hip: 16.5 clk per wave-level load), so a three-channel gather spends
// ~200 clk of TCP time per wave on delivery alone -- as much as its HBM time -- and more when the
// displacement grows (profiles/r03_window_gather.md).  For a smooth displacement the sources of a compact
// tile of voxels are a compact block too: tile + halo, translated by the tile's mean displacement.  The
// kernels here move that block global -> LDS with `buffer_load_dwordx4 ... lds` (62 B/clk, no VGPR round
// trip) in coalesced 16-byte chunks and take the corners from LDS (ds_read2_b32 on the same (zb, zb+1)
// pair the pair-gather uses, same selects, same value expression -> bit-identical results).
//
// Geometry: a workgroup of NT = 512*XS threads owns TX x TY x TZ = 8 x 16 x 32 voxels (lane = z, 16 rows
// of y, XS x-slabs, U = 8/XS voxels per lane along x).  The window is WX x WY = 13 x 21 rows of WZ = 44
// floats (11 chunks), i.e. the tile plus a halo of 2 voxels in x / y and at least 4 in z around the
// position the tile's centre voxel is displaced to.  One channel's window is 48 KB: two workgroups per CU.
// Samples with a corner outside the window take their corners with the pair gathers (per lane, wave-uniform skip).
#pragma once
#include "common.hpp"

namespace lago {

struct GW {
#ifndef LAGO_GW_TY
#define LAGO_GW_TY 16
#endif
    static constexpr int TX = 8, TY = LAGO_GW_TY, TZ = 32, H = 2, HZ = 4;
    static constexpr int NT = 32 * TY;   // threads of a workgroup: lane = z, TY rows of y, one x slab, TX voxels per lane along x
    static constexpr int WX = TX + 1 + 2 * H, WY = TY + 1 + 2 * H, WZC = 11, WZ = 4 * WZC;
    static constexpr int NCHUNK = WX * WY * WZC;
    static constexpr uint32_t kOutside = 0x80000000u;  // beyond any plane (nvox * 8 < 2^32): reads 0
    template <int NT>
    static constexpr int rounds() { return (NCHUNK + NT - 1) / NT; }
    template <int NT>
    static constexpr size_t lds_bytes() { return (size_t)rounds<NT>() * NT * 16; }
};

// launch decomposition: tiles of one batch item, z fastest
struct GWGrid {
    uint32_t ntx, nty, ntz, per_item, total;
    FastDiv d_item, d_tz, d_ty;
};

// nz % 4 == 0 keeps every row 16-byte aligned relative to the plane; partially filled tiles idle lanes, so
// shapes that fill less than 85 % of their tiles stay on the pair-gather kernels.
inline bool make_gwgrid(GWGrid &w, const Geom &g, int64_t nn) {
    if (g.nz < 4 || (g.nz & 3) || g.nx < 2 || g.ny < 2) return false;
    w.ntx = (g.nx + GW::TX - 1) / GW::TX;
    w.nty = (g.ny + GW::TY - 1) / GW::TY;
    w.ntz = (g.nz + GW::TZ - 1) / GW::TZ;
    w.per_item = w.ntx * w.nty * w.ntz;
    const uint64_t total = (uint64_t)w.per_item * (uint64_t)nn;
    if (total == 0 || total >= (1ull << 31)) return false;
    if ((double)g.nvox < 0.85 * (double)w.per_item * (GW::TX * GW::TY * GW::TZ)) return false;
    w.total = (uint32_t)total;
    w.d_item = FastDiv(w.per_item);
    w.d_tz = FastDiv(w.ntz);
    w.d_ty = FastDiv(w.nty);
    return true;
}

struct GWTile {
    uint32_t n;
    int x0, y0, z0;
};
__device__ __forceinline__ GWTile gw_tile(const GWGrid &w, int rev) {
    const uint32_t L = block_order(blockIdx.x, w.total, rev);
    GWTile t;
    t.n = w.d_item.div(L);
    const uint32_t tb = L - t.n * w.per_item;
    const uint32_t txy = w.d_tz.div(tb);
    const uint32_t tx = w.d_ty.div(txy);
    t.z0 = (int)(tb - txy * w.ntz) * GW::TZ;
    t.y0 = (int)(txy - tx * w.nty) * GW::TY;
    t.x0 = (int)tx * GW::TX;
    return t;
}

// Window origin from the position the tile's centre voxel is displaced to (any uniform choice is correct:
// samples outside the window use the pair gathers).
struct GWOrigin {
    int ox, oy, oz;
};
__device__ __forceinline__ GWOrigin gw_origin(const GWTile &t, const Geom &g, int px, int py, int pz) {
    // (px, py, pz): floor of the displaced position of voxel (cxi, cyi, czi) below
    const int cxi = min(t.x0 + GW::TX / 2, g.nx - 1), cyi = min(t.y0 + GW::TY / 2, g.ny - 1),
              czi = min(t.z0 + GW::TZ / 2, g.nz - 1);
    GWOrigin o;
    o.ox = t.x0 + (px - cxi) - GW::H;
    o.oy = t.y0 + (py - cyi) - GW::H;
    o.oz = (t.z0 + (pz - czi) - GW::HZ) & ~3;  // floor to a chunk boundary
    return o;
}

// The 16-byte chunks of the window this lane moves (the same for every channel): byte offsets into a plane.
template <int NT>
struct GWLoader {
    uint32_t src[GW::rounds<NT>()];
    __device__ __forceinline__ void plan(const GWOrigin &o, const Geom &g) {
#pragma unroll
        for (int r = 0; r < GW::rounds<NT>(); ++r) {
            const uint32_t c = (uint32_t)(r * NT) + threadIdx.x;
            const uint32_t row = c / (uint32_t)GW::WZC, cz = c - row * GW::WZC;
            const uint32_t wx = row / (uint32_t)GW::WY, wy = row - wx * GW::WY;
            const int gx = o.ox + (int)wx, gy = o.oy + (int)wy, gz = o.oz + 4 * (int)cz;
            const bool ok = c < (uint32_t)GW::NCHUNK && (uint32_t)gx < (uint32_t)g.nx && (uint32_t)gy < (uint32_t)g.ny &&
                            (uint32_t)gz < (uint32_t)g.nz;
            src[r] = ok ? (((uint32_t)gx * (uint32_t)g.ny + (uint32_t)gy) * (uint32_t)g.nz + (uint32_t)gz) * 4u : GW::kOutside;
        }
    }
    // `plane` is wave-uniform; chunk c lands at win + 4 c floats (lane l of a wave at the wave's base + 16 l bytes)
    __device__ __forceinline__ void issue(const float *__restrict__ plane, uint32_t plane_bytes, float *win) const {
        const BufRsrc r = make_rsrc(plane, plane_bytes);
#pragma unroll
        for (int q = 0; q < GW::rounds<NT>(); ++q) {
            float *dst = win + (size_t)(q * NT + (threadIdx.x & ~63u)) * 4;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)dst, 16, src[q], 0, 0, 0);
        }
    }
};

// One sample: the first lines of Lerp3::setup (same expressions), then window indices instead of byte offsets.
struct GWLerp {
    float t, u, v;
    uint32_t pk;  // window index of (fx, fy, zb) | dX << 14 | dY << 15 | f_hi << 16 | c_lo << 17
    // returns false if a corner lies outside the window
    __device__ __forceinline__ bool setup(float x, float y, float z, const Geom &g, const GWOrigin &o) {
        const int flx = lg_floor(x), fly = lg_floor(y), flz = lg_floor(z);
        t = x - (float)flx;
        u = y - (float)fly;
        v = z - (float)flz;
        const int fx = clamp1(flx, g.nx), cx = clamp1(flx + 1, g.nx);
        const int fy = clamp1(fly, g.ny), cy = clamp1(fly + 1, g.ny);
        const int fz = clamp1(flz, g.nz), cz = clamp1(flz + 1, g.nz);
        const int zb = min(fz, g.nz - 2);
        const bool f_hi = fz != zb, c_lo = cz == zb;
        const int ax = fx - o.ox, ay = fy - o.oy, az = zb - o.oz;
        // branch-free: (unsigned)a <= (unsigned)b folds the two-sided tests (fx <= cx <= fx + 1)
        const bool in = ((uint32_t)ax < (uint32_t)(GW::WX - (cx - fx))) & ((uint32_t)ay < (uint32_t)(GW::WY - (cy - fy))) &
                        ((uint32_t)az < (uint32_t)(GW::WZ - 1));
        const uint32_t wi = in ? (uint32_t)((ax * GW::WY + ay) * GW::WZ + az) : 0u;
        pk = wi | (cx != fx ? 1u << 14 : 0u) | (cy != fy ? 1u << 15 : 0u) | (f_hi ? 1u << 16 : 0u) | (c_lo ? 1u << 17 : 0u);
        // pin the packed word here: otherwise its ingredients (six coordinates per sample) are kept alive up to the
        // first use behind the barrier
        asm volatile("" : "+v"(pk), "+v"(t), "+v"(u), "+v"(v));
        return in;
    }
    // the eight corners in Lerp3's order (rows (fx,fy) (cx,fy) (cx,cy) (fx,cy) at floor z, then at ceil z)
    __device__ __forceinline__ void fetch(const float *win, float (&c)[8]) const {
        uint32_t p = pk;
        asm volatile("" : "+v"(p));  // rebuild the four addresses per channel; keeping them costs 4 VGPRs per sample
        const uint32_t b0 = p & 0x3fffu;
        const uint32_t dX = (p & (1u << 14)) ? (uint32_t)(GW::WY * GW::WZ) : 0u, dY = (p & (1u << 15)) ? (uint32_t)GW::WZ : 0u;
        const bool f_hi = p & (1u << 16), c_lo = p & (1u << 17);
        const uint32_t rb[4] = {b0, b0 + dX, b0 + dX + dY, b0 + dY};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float lo = win[rb[q]], hi = win[rb[q] + 1];
            c[q] = f_hi ? hi : lo;
            c[q + 4] = c_lo ? lo : hi;
        }
    }
    // fetch + value_of with the z-border selection behind the two (x, y) interpolations (Lerp3::value, common.hpp: same bits,
    // two selects instead of eight)
    __device__ __forceinline__ float value(const float *win) const {
        uint32_t p = pk;
        asm volatile("" : "+v"(p));
        const uint32_t b0 = p & 0x3fffu;
        const uint32_t dX = (p & (1u << 14)) ? (uint32_t)(GW::WY * GW::WZ) : 0u, dY = (p & (1u << 15)) ? (uint32_t)GW::WZ : 0u;
        const bool f_hi = p & (1u << 16), c_lo = p & (1u << 17);
        const uint32_t rb[4] = {b0, b0 + dX, b0 + dX + dY, b0 + dY};
        float lo[4], hi[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { lo[q] = win[rb[q]]; hi[q] = win[rb[q] + 1]; }
        const float omt = 1.f - t, omu = 1.f - u, omv = 1.f - v;
        const float glo = lg_fma(omu, lg_fma(omt, lo[0], t * lo[1]), u * lg_fma(omt, lo[3], t * lo[2]));
        const float ghi = lg_fma(omu, lg_fma(omt, hi[0], t * hi[1]), u * lg_fma(omt, hi[3], t * hi[2]));
        return lg_fma(omv, f_hi ? ghi : glo, v * (c_lo ? glo : ghi));
    }
    // Lerp3::value_of (include/interp.h:115-122)
    __device__ __forceinline__ float value_of(const float (&c)[8]) const {
        const float omt = 1.f - t, omu = 1.f - u, omv = 1.f - v;
        return lg_fma(omv, lg_fma(omu, lg_fma(omt, c[0], t * c[1]), u * lg_fma(omt, c[3], t * c[2])),
                      v * lg_fma(omu, lg_fma(omt, c[4], t * c[5]), u * lg_fma(omt, c[7], t * c[6])));
    }
};

}  // namespace lago
