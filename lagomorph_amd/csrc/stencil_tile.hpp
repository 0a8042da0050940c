// LDS-staged row tiles with a one-voxel halo for the clamped central-difference stencils (gfx950).
//
// What this replaces: the reference evaluates the Jacobian terms voxel by voxel from global memory
// (cuda/diff.cu:63-127 over grad_point, include/diff.h:55-76); the first HIP kernels did the same with one lane per
// voxel -- six neighbour dwords per differentiated component through the CU's vector-memory path.  That path, not
// HBM, bounds the gather / stencil kernels of this library (profiles/r02_gather_experiments.md: about 32 bytes per
// clock and CU into the vector registers, however the dwords are fetched), so the stencil operand is staged ONCE per
// workgroup instead:
//
//   * a workgroup owns a tile of TX x TY z-rows (each row = all nz voxels of one (x, y)) of one batch item;
//   * the differentiated planes of the tile are written to LDS -- the centre values from the registers that hold
//     them anyway, the 2 TX + 2 TY halo rows by coalesced row loads (one wave per halo row, so the row decode is
//     scalar), the two z ends of every row replicated -- with every row taken at CLAMPED grid coordinates:
//     LDS row (a, b) holds grid row (clamp(x0 + a), clamp(y0 + b)), so `f[+1] - f[-1]` on the tile IS the reference's
//     clamped central difference (a missing neighbour is the centre value itself, include/extrap.h:41-44), and a
//     ragged last tile needs no special case;
//   * after ONE barrier every lane reads its six neighbours per component with ds_read (LDS delivers 128 B per clock
//     and CU and is idle in these kernels).
//
// Loaded dwords per voxel of the stencil operand: 1 + (2 TX + 2 TY) / (TX TY) instead of 7 (2.5 for a 2 x 4 tile).
#pragma once

#include "common.hpp"

namespace lago {

struct RowTile {
    int TX, TY;            // rows of the tile along x and y
    int RY, P;             // LDS rows per x layer (TY + 2), row pitch in elements (nz + 2)
    uint32_t ntx, nty, tiles_per_item, total;
    uint32_t tile_vox;     // TX * TY * nz
    uint32_t nhrows;       // halo rows per plane: 2 TY (x faces) + 2 TX (y faces)
    uint32_t plane;        // LDS elements per staged plane: (TX + 2) * RY * P
    FastDiv d_tiles, d_nty, d_nz, d_TY, d_nhrows;
};

// Tile for workgroups that cover `cap` voxels (threads x voxels per thread): as many whole z-rows as fit, arranged
// TX x TY with the smallest halo; at least 2 x 2 rows.  `nplanes` planes of `esize`-byte elements are staged and
// at most `max_halo_rows_per_wave` halo rows may fall to one wave (compile-time unrolling of the loader).
inline bool make_row_tile(RowTile &t, const Geom &g, int64_t nn, int cap, int nthreads, int nplanes, int esize,
                          int max_halo_rows_per_wave, size_t &smem) {
    if (g.nz < 2 || g.nx < 2 || g.ny < 2) return false;
    const int rows = cap / g.nz;
    if (rows < 4) return false;
    const int waves = nthreads / 64;
    int bx = 0, by = 0;
    double best = 1e30;
    for (int tx = 2; tx <= rows && tx <= g.nx; ++tx) {
        for (int ty = 2; ty <= rows / tx && ty <= g.ny; ++ty) {
            // the loader's compile-time bound on halo rows per wave
            if ((int)(((2 * tx + 2 * ty) * nplanes + waves - 1) / waves) > max_halo_rows_per_wave) continue;
            // halo rows per tile row, plus the lanes a partly filled workgroup leaves idle
            const double cost = (2.0 * tx + 2.0 * ty) / (tx * ty) + 4.0 * (1.0 - (double)tx * ty * g.nz / cap);
            if (cost < best) { best = cost; bx = tx; by = ty; }
        }
    }
    if (!bx || (int64_t)bx * by * g.nz * 2 < cap) return false;  // short rows: a workgroup would be half empty
    t.TX = bx; t.TY = by;
    t.RY = by + 2;
    t.P = g.nz + 2;
    t.nhrows = 2u * by + 2u * bx;
    t.plane = (uint32_t)(bx + 2) * t.RY * t.P;
    smem = (size_t)t.plane * nplanes * esize;
    if (smem > 64 * 1024) return false;
    t.ntx = (g.nx + bx - 1) / bx;
    t.nty = (g.ny + by - 1) / by;
    t.tiles_per_item = t.ntx * t.nty;
    const int64_t total = (int64_t)t.tiles_per_item * nn;
    if (total >= (1ll << 31)) return false;
    t.total = (uint32_t)total;
    t.tile_vox = (uint32_t)bx * by * g.nz;
    t.d_tiles = FastDiv(t.tiles_per_item);
    t.d_nty = FastDiv(t.nty);
    t.d_nz = FastDiv((uint32_t)g.nz);
    t.d_TY = FastDiv((uint32_t)by);
    t.d_nhrows = FastDiv(t.nhrows);
    return true;
}

// One voxel of the tile as a lane sees it.
struct TileVox {
    uint32_t s;    // flattened grid index of the (clamped) voxel: always a valid address
    uint32_t li;   // its element index inside a staged plane
    int i, j, k;   // clamped grid coordinates
    bool has;      // the lane has a slot in the tile (tile_vox need not fill the workgroup)
    bool ok;       // ... and the voxel lies inside the grid (ragged last tiles): results are stored
};

__device__ __forceinline__ TileVox tile_voxel(const RowTile &t, const Geom &g, int x0, int y0, uint32_t v) {
    TileVox q;
    q.has = v < t.tile_vox;
    const uint32_t vv = q.has ? v : 0u;
    const uint32_t row = t.d_nz.div(vv);
    const uint32_t k = vv - row * (uint32_t)g.nz;
    const uint32_t a = t.d_TY.div(row);
    const uint32_t b = row - a * (uint32_t)t.TY;
    q.ok = q.has && x0 + (int)a < g.nx && y0 + (int)b < g.ny;
    q.i = min(x0 + (int)a, g.nx - 1);
    q.j = min(y0 + (int)b, g.ny - 1);
    q.k = (int)k;
    q.s = ((uint32_t)q.i * (uint32_t)g.ny + (uint32_t)q.j) * (uint32_t)g.nz + k;
    q.li = ((a + 1u) * (uint32_t)t.RY + (b + 1u)) * (uint32_t)t.P + k + 1u;
    return q;
}

// A lane's own centre value into the staged plane, the z ends of its row replicated.
template <typename R>
__device__ __forceinline__ void tile_put(R *plane, const TileVox &q, int nz, R val) {
    if (!q.has) return;
    plane[q.li] = val;
    if (q.k == 0) plane[q.li - 1] = val;
    if (q.k == nz - 1) plane[q.li + 1] = val;
}

// The halo rows of NPL planes (plane p of the batch item at base + p * nv; staged at lds + p * t.plane).  Wave w takes
// halo rows w, w + NW, ... of the NPL * nhrows rows; its lanes run along z in ZC chunks of 64.  RI = rows per wave,
// ZC = ceil(nz / 64) (both compile-time bounds; the host checks).  Two steps, so that a kernel can put other loads in
// flight between them: issue() requests every row into registers, commit() writes them to LDS.
template <typename R, int NT, int NPL, int RI, int ZC>
struct TileHalo {
    R val[RI][ZC];
    uint32_t dst[RI];
    __device__ __forceinline__ void issue(const R *__restrict__ base, size_t nv, const RowTile &t, const Geom &g, int x0,
                                          int y0) {
        constexpr int NW = NT / 64;
        const int lane = threadIdx.x & 63;
        const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const uint32_t nrows = t.nhrows * NPL;
#pragma unroll
        for (int it = 0; it < RI; ++it) {
            const uint32_t hr = wave + (uint32_t)it * NW;   // wave-uniform
            dst[it] = 0xffffffffu;
            if (hr < nrows) {
                const uint32_t p = t.d_nhrows.div(hr);
                const uint32_t h = hr - p * t.nhrows;
                int a, b;
                if (h < 2u * (uint32_t)t.TY) {   // x faces: a = -1 or TX
                    a = h < (uint32_t)t.TY ? -1 : t.TX;
                    b = (int)(h < (uint32_t)t.TY ? h : h - (uint32_t)t.TY);
                } else {                          // y faces: b = -1 or TY
                    const uint32_t h2 = h - 2u * (uint32_t)t.TY;
                    b = h2 < (uint32_t)t.TX ? -1 : t.TY;
                    a = (int)(h2 < (uint32_t)t.TX ? h2 : h2 - (uint32_t)t.TX);
                }
                const int gi = max(0, min(x0 + a, g.nx - 1)), gj = max(0, min(y0 + b, g.ny - 1));
                const R *src = base + (size_t)p * nv + ((size_t)gi * g.ny + gj) * g.nz;
                dst[it] = p * t.plane + ((uint32_t)(a + 1) * (uint32_t)t.RY + (uint32_t)(b + 1)) * (uint32_t)t.P + 1u;
#pragma unroll
                for (int z = 0; z < ZC; ++z) {
                    const int k = lane + 64 * z;
                    val[it][z] = k < g.nz ? src[k] : (R)0;
                }
            }
        }
    }
    __device__ __forceinline__ void commit(R *lds, const Geom &g) const {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int it = 0; it < RI; ++it) {
            if (dst[it] != 0xffffffffu) {
#pragma unroll
                for (int z = 0; z < ZC; ++z) {
                    const int k = lane + 64 * z;
                    if (k < g.nz) lds[dst[it] + (uint32_t)k] = val[it][z];
                }
            }
        }
    }
};

}  // namespace lago
