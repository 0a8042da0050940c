// LDS-staged trilinear gathers for 3D fields -- gfx950.
//
// interp_forward (cuda/interp.cu:48-78), deform.compose (deform.py:53-55) and adjrep.Ad_star
// (adjrep.py:86-97) all sample an image at x + dt*u(x).  Straight from global memory that is four
// 8-byte pair loads per voxel-channel, and the L1 (TCP) is what they saturate: PMC counters on the
// direct kernels show ~4.3 L1 accesses per voxel (0.67 per clock per CU) at an 89 % hit rate, with
// HBM traffic equal to the algorithmic bytes -- the gather is bound by L1 request throughput, not by
// HBM.  Here a workgroup owns a TX x TY x TZ tile of output voxels and, per channel, stages the
// part of the image its samples fall into -- a window placed at the tile origin plus the
// displacement probed at the tile centre -- in LDS with coalesced row loads (about 2.3 loaded cells
// per voxel instead of 8 gathered ones), then every lane reads its 2x2x2 corners from LDS
// (4 x ds_read2_b32).  Samples whose footprint is clamped at a border or leaves the window take the
// direct global-memory path, so any displacement is correct and smooth ones are fast.
// The arithmetic is that of Lerp3 (common.hpp), so results are bit-identical to the direct kernels.
#include "common.hpp"

namespace lago {

enum { GOP_INTERP = 0, GOP_COMPOSE = 1, GOP_ADSTAR = 2 };

struct GTile {
    int nx, ny, nz;
    int TX, TY, TZ;          // tile of output voxels
    int WX, WY, WZ;          // window extents (cells)
    int MX, MY, MZ;          // margin below the probed origin
    uint32_t ntx, nty, ntz, tiles_per_item, total, tile_voxels;
    FastDiv d_tiles, d_tyz, d_tz, d_TyTz, d_Tz, d_wey, d_wez;
};

// include/interp.h:115-122 on eight corner values (the expression of Lerp3::value)
template <typename R>
__device__ __forceinline__ R lerp8(const R (&c)[8], R t, R u, R v) {
    const R omt = (R)1.f - t, omu = (R)1.f - u, omv = (R)1.f - v;
    return lg_fma(omv, lg_fma(omu, lg_fma(omt, c[0], t * c[1]), u * lg_fma(omt, c[3], t * c[2])),
                  v * lg_fma(omu, lg_fma(omt, c[4], t * c[5]), u * lg_fma(omt, c[7], t * c[6])));
}

// Direct global-memory sample for footprints that leave the window or are clamped at a border.
// Deliberately not inlined: it is the rare path, and inlining it VPT times tripled the register count.
template <typename R>
__device__ __forceinline__ R direct_sample(const R *__restrict__ plane, R x, R y, R z, int nx, int ny, int nz) {
    Lerp3<R, true> L;
    L.setup(x, y, z, nx, ny, nz);
    return L.value(plane);
}

template <typename R, int OP, bool BC, bool UNIT, int NT, int VPT>
__global__ __launch_bounds__(NT) void gather_tiled_kernel(R *__restrict__ out, const R *__restrict__ img,
                                                          const R *__restrict__ u, double ds, double dt, int nc,
                                                          GTile tg) {
    extern __shared__ __align__(16) unsigned char lago_smem[];
    R *win = reinterpret_cast<R *>(lago_smem);
    const int nx = tg.nx, ny = tg.ny, nz = tg.nz;
    const size_t nv = (size_t)nx * ny * nz;

    const uint32_t Lb = xcd_swizzle(blockIdx.x, tg.total);
    const uint32_t n = tg.d_tiles.div(Lb);
    uint32_t r = Lb - n * tg.tiles_per_item;
    const uint32_t bx = tg.d_tyz.div(r);
    r -= bx * (tg.nty * tg.ntz);
    const uint32_t by = tg.d_tz.div(r);
    const uint32_t bz = r - by * tg.ntz;
    const int x0 = bx * tg.TX, y0 = by * tg.TY, z0 = bz * tg.TZ;
    const int ex = min(tg.TX, nx - x0), ey = min(tg.TY, ny - y0), ez = min(tg.TZ, nz - z0);

    const R *un = u + (size_t)n * 3 * nv;
    const R *imn = BC ? img : img + (size_t)n * nc * nv;
    R *on = out + (size_t)n * nc * nv;

    // window origin: tile origin + displacement probed at the tile centre - margin (speed only)
    int wx0, wy0, wz0;
    {
        const size_t sc = ((size_t)(x0 + ex / 2) * ny + (y0 + ey / 2)) * nz + (z0 + ez / 2);
        const float fds = (float)ds;
        wx0 = x0 + (int)floorf(fds * (float)un[sc]) - tg.MX;
        wy0 = y0 + (int)floorf(fds * (float)un[sc + nv]) - tg.MY;
        wz0 = z0 + (int)floorf(fds * (float)un[sc + 2 * nv]) - tg.MZ;
    }
    const int wex = min(tg.WX, nx), wey = min(tg.WY, ny), wez = min(tg.WZ, nz);
    wx0 = max(0, min(wx0, nx - wex));
    wy0 = max(0, min(wy0, ny - wey));
    wz0 = max(0, min(wz0, nz - wez));
    const int WY = tg.WY, WZ = tg.WZ;
    const int wbase = -((wx0 * WY + wy0) * WZ + wz0);

    // per-voxel state: position, fractions, window cell of the floor corner (or -1: direct path)
    uint32_t s[VPT];
    bool live[VPT];
    R uu[3][VPT], ft[VPT], fu[VPT], fv[VPT];
    int cell[VPT];
    // tile-local voxel id -> grid coordinates (dead lanes read the tile origin)
    auto coords = [&](int e, int &i, int &j, int &k) -> bool {
        const uint32_t t = threadIdx.x + e * NT;
        const uint32_t a = tg.d_TyTz.div(t);
        const uint32_t rr = t - a * (uint32_t)(tg.TY * tg.TZ);
        const uint32_t b = tg.d_Tz.div(rr);
        const uint32_t cc = rr - b * (uint32_t)tg.TZ;
        const bool ok = t < tg.tile_voxels && (int)a < ex && (int)b < ey && (int)cc < ez;
        i = ok ? x0 + (int)a : x0;
        j = ok ? y0 + (int)b : y0;
        k = ok ? z0 + (int)cc : z0;
        return ok;
    };
#pragma unroll
    for (int e = 0; e < VPT; ++e) {
        int i, j, k;
        live[e] = coords(e, i, j, k);
        s[e] = ((uint32_t)i * ny + j) * nz + k;
#pragma unroll
        for (int d = 0; d < 3; ++d) uu[d][e] = un[(size_t)d * nv + s[e]];
    }
#pragma unroll
    for (int e = 0; e < VPT; ++e) {
        int i, j, k;
        coords(e, i, j, k);
        const R hx = sample_pos_t<R, UNIT>(i, ds, uu[0][e]);
        const R hy = sample_pos_t<R, UNIT>(j, ds, uu[1][e]);
        const R hz = sample_pos_t<R, UNIT>(k, ds, uu[2][e]);
        const int fx = lg_floor(hx), fy = lg_floor(hy), fz = lg_floor(hz);
        ft[e] = hx - (R)fx;
        fu[e] = hy - (R)fy;
        fv[e] = hz - (R)fz;
        // unclamped 2x2x2 footprint inside the window (the window lies inside the grid)
        const bool inside = (unsigned)(fx - wx0) < (unsigned)(wex - 1) && (unsigned)(fy - wy0) < (unsigned)(wey - 1) &&
                            (unsigned)(fz - wz0) < (unsigned)(wez - 1);
        cell[e] = inside ? (fx * WY + fy) * WZ + fz + wbase : -1;
    }

    R acc[OP == GOP_ADSTAR ? 3 : 1][VPT];
    const uint32_t ncells = (uint32_t)(wex * wey * wez);

    for (int c = 0; c < nc; ++c) {
        const R *plane = imn + (size_t)c * nv;
        if (c) __syncthreads();  // everyone is done reading the previous channel's window
        // stage the window: cells flattened over the workgroup (lanes run along z, rows follow each
        // other), FK independent loads in flight per thread before the first LDS store
        constexpr int FK = 8;
        for (uint32_t f0 = threadIdx.x; f0 < ncells; f0 += NT * FK) {
            R tmp[FK];
            uint32_t dst[FK];
#pragma unroll
            for (int q = 0; q < FK; ++q) {
                const uint32_t f = f0 + q * NT;
                const uint32_t fc = f < ncells ? f : 0;
                const uint32_t row = tg.d_wez.div(fc), lz = fc - row * (uint32_t)wez;
                const uint32_t lx = tg.d_wey.div(row), ly = row - lx * (uint32_t)wey;
                dst[q] = f < ncells ? (lx * (uint32_t)WY + ly) * (uint32_t)WZ + lz : 0xffffffffu;
                tmp[q] = plane[((size_t)(wx0 + lx) * ny + (wy0 + ly)) * nz + (wz0 + lz)];
            }
#pragma unroll
            for (int q = 0; q < FK; ++q)
                if (dst[q] != 0xffffffffu) win[dst[q]] = tmp[q];
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < VPT; ++e) {
            R val;
            if (cell[e] >= 0) {
                R cv[8];
                const R *w0 = win + cell[e];
                const R *w1 = w0 + WY * WZ;
                cv[0] = w0[0]; cv[4] = w0[1];
                cv[1] = w1[0]; cv[5] = w1[1];
                cv[2] = w1[WZ]; cv[6] = w1[WZ + 1];
                cv[3] = w0[WZ]; cv[7] = w0[WZ + 1];
                val = lerp8(cv, ft[e], fu[e], fv[e]);
            } else {
                int i, j, k;
                coords(e, i, j, k);
                val = direct_sample<R>(plane, sample_pos_t<R, UNIT>(i, ds, uu[0][e]), sample_pos_t<R, UNIT>(j, ds, uu[1][e]),
                                       sample_pos_t<R, UNIT>(k, ds, uu[2][e]), nx, ny, nz);
            }
            if ((e & 1) == 1) __builtin_amdgcn_sched_barrier(0);  // two voxels' corner reads in flight at a time
            if (OP == GOP_INTERP) {
                if (live[e]) on[(size_t)c * nv + s[e]] = val;
            } else if (OP == GOP_COMPOSE) {
                const R a = (R)ds * uu[c][e];  // torch multiplies by the scalar rounded to the tensor dtype
                const R b = (R)dt * val;
                if (live[e]) on[(size_t)c * nv + s[e]] = a + b;
            } else {
                acc[c][e] = val;
            }
        }
    }

    if (OP == GOP_ADSTAR) {
        // (D phiinv + I) applied to the resampled momentum: the expression of jtv_fwd_kernel (diff.hip)
        const int syz = ny * nz;
#pragma unroll
        for (int e = 0; e < VPT; ++e) {
            int i, j, k;
            coords(e, i, j, k);
            const int px = i + 1 < nx ? syz : 0, mx = i > 0 ? -syz : 0;
            const int py = j + 1 < ny ? nz : 0, my = j > 0 ? -nz : 0;
            const int pz = k + 1 < nz ? 1 : 0, mz = k > 0 ? -1 : 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const R *pc = un + (size_t)c * nv + s[e];
                R gq[3];
                gq[0] = (R)0.5f * (pc[px] - pc[mx]);
                gq[1] = (R)0.5f * (pc[py] - pc[my]);
                gq[2] = (R)0.5f * (pc[pz] - pc[mz]);
                gq[c] = gq[c] + (R)1.0;
                const R sacc = lg_fma(gq[2], acc[2 % (OP == GOP_ADSTAR ? 3 : 1)][e],
                                      lg_fma(gq[0], acc[0][e], gq[1] * acc[1 % (OP == GOP_ADSTAR ? 3 : 1)][e]));
                if (live[e]) on[(size_t)c * nv + s[e]] = sacc;
            }
        }
    }
}

// ---- host side -----------------------------------------------------------------------------------

// 0 (default): direct gathers; 1: LDS-staged gathers where the shape allows.  Measured on MI355X
// (tools/sweep_gather.py, batch 8 x 3 x 128^3, smooth 4-voxel displacement): compose 183 us direct
// against 284 us staged (8 x 8 x 64 tile, margin 1; 382 us at margin 2), ad_star 257 vs 328,
// interp C=1 86 vs 127.  The staged kernel is correct for any displacement and bit-identical, but
// its per-channel barrier phases (fill, read) do not overlap inside a workgroup and only two
// 1024-thread workgroups fit a CU, while the direct kernel keeps 28 independent waves per CU in
// flight; it is kept as an opt-in for fields rough enough to defeat the L1.
int g_gather_mode = 0;
static int g_gtile_cfg[7] = {8, 8, 64, 1, 1, 1, 1024};   // TX TY TZ, margins MX MY MZ, threads

static bool make_gtile(GTile &tg, const Geom &g, int64_t nn, size_t elem, size_t &smem, int &nt, int &vpt) {
    int TX = g_gtile_cfg[0], TY = g_gtile_cfg[1], TZ = g_gtile_cfg[2];
    const int MX = g_gtile_cfg[3], MY = g_gtile_cfg[4], MZ = g_gtile_cfg[5];
    nt = g_gtile_cfg[6] >= 1024 ? 1024 : (g_gtile_cfg[6] >= 512 ? 512 : 256);
    if (TX < 1 || TY < 1 || TZ < 1 || MX < 0 || MY < 0 || MZ < 0) return false;
    TX = TX < g.nx ? TX : g.nx;
    TY = TY < g.ny ? TY : g.ny;
    TZ = TZ < g.nz ? TZ : g.nz;
    const int tv = TX * TY * TZ;
    if (tv < 4 * nt) return false;  // small volumes stay with the direct kernels
    vpt = (tv + nt - 1) / nt;
    if (vpt != 4 && vpt != 8) return false;
    tg.nx = g.nx; tg.ny = g.ny; tg.nz = g.nz;
    tg.TX = TX; tg.TY = TY; tg.TZ = TZ;
    tg.MX = MX; tg.MY = MY; tg.MZ = MZ;
    tg.WX = TX + 1 + 2 * MX; tg.WY = TY + 1 + 2 * MY; tg.WZ = TZ + 1 + 2 * MZ;
    if (tg.WX > g.nx) tg.WX = g.nx;
    if (tg.WY > g.ny) tg.WY = g.ny;
    if (tg.WZ > g.nz) tg.WZ = g.nz;
    smem = (size_t)tg.WX * tg.WY * tg.WZ * elem;
    if (smem > 64 * 1024) return false;
    tg.ntx = (g.nx + TX - 1) / TX;
    tg.nty = (g.ny + TY - 1) / TY;
    tg.ntz = (g.nz + TZ - 1) / TZ;
    tg.tiles_per_item = tg.ntx * tg.nty * tg.ntz;
    const int64_t total = (int64_t)tg.tiles_per_item * nn;
    if (total >= (1ll << 31)) return false;
    tg.total = (uint32_t)total;
    tg.tile_voxels = (uint32_t)tv;
    tg.d_tiles = FastDiv(tg.tiles_per_item);
    tg.d_tyz = FastDiv(tg.nty * tg.ntz);
    tg.d_tz = FastDiv(tg.ntz);
    tg.d_TyTz = FastDiv((uint32_t)(TY * TZ));
    tg.d_Tz = FastDiv((uint32_t)TZ);
    tg.d_wey = FastDiv((uint32_t)tg.WY);
    tg.d_wez = FastDiv((uint32_t)tg.WZ);
    return true;
}

template <typename R, int OP, bool BC, bool UNIT>
static void launch_gtile(R *out, const R *img, const R *u, double ds, double dt, int nc, const GTile &tg, size_t smem,
                         int nt, int vpt, hipStream_t s) {
#define GO(NT_, VPT_)                                                                                              \
    hipLaunchKernelGGL((gather_tiled_kernel<R, OP, BC, UNIT, NT_, VPT_>), dim3(tg.total), dim3(NT_), smem, s, out, img, \
                       u, ds, dt, nc, tg)
    if (nt == 1024) {
        if (vpt == 4) GO(1024, 4); else GO(1024, 8);
    } else if (nt == 512) {
        if (vpt == 4) GO(512, 4); else GO(512, 8);
    } else {
        if (vpt == 4) GO(256, 4); else GO(256, 8);
    }
#undef GO
}

// Returns LAGO_OK, or 1 when the shape is left to the direct kernels.
template <typename R>
int gather_tiled(int op, R *out, const R *img, const R *u, double ds, double dt, int nc, int64_t nn, const Geom &g,
                 bool bc, hipStream_t s) {
    if (!g_gather_mode || g.nz < 2) return 1;
    GTile tg;
    size_t smem;
    int nt, vpt;
    if (!make_gtile(tg, g, nn, sizeof(R), smem, nt, vpt)) return 1;
    const bool unit = unit_dt<R>(ds);
#define BY_UNIT(OP_, BC_)                                                                        \
    do {                                                                                         \
        if (unit) launch_gtile<R, OP_, BC_, true>(out, img, u, ds, dt, nc, tg, smem, nt, vpt, s); \
        else launch_gtile<R, OP_, BC_, false>(out, img, u, ds, dt, nc, tg, smem, nt, vpt, s);     \
    } while (0)
    if (op == GOP_INTERP) {
        if (bc) BY_UNIT(GOP_INTERP, true); else BY_UNIT(GOP_INTERP, false);
    } else if (op == GOP_COMPOSE) {
        BY_UNIT(GOP_COMPOSE, false);
    } else {
        launch_gtile<R, GOP_ADSTAR, false, true>(out, img, u, 1.0, 1.0, 3, tg, smem, nt, vpt, s);
    }
#undef BY_UNIT
    return LAGO_OK;
}

template int gather_tiled<float>(int, float *, const float *, const float *, double, double, int, int64_t, const Geom &,
                                 bool, hipStream_t);
template int gather_tiled<double>(int, double *, const double *, const double *, double, double, int, int64_t,
                                  const Geom &, bool, hipStream_t);

}  // namespace lago

extern "C" {
// 0 (default): direct gathers; 1: LDS-staged gather kernels for 3D interp_forward / compose / ad_star.
void lago_set_gather_mode(int mode) { lago::g_gather_mode = mode ? 1 : 0; }
// Tuning hook: tile TX TY TZ, window margins MX MY MZ, threads per workgroup (256 / 512).  Speed only.
void lago_set_gather_tile(int tx, int ty, int tz, int mx, int my, int mz, int nthreads) {
    lago::g_gtile_cfg[0] = tx; lago::g_gtile_cfg[1] = ty; lago::g_gtile_cfg[2] = tz;
    lago::g_gtile_cfg[3] = mx; lago::g_gtile_cfg[4] = my; lago::g_gtile_cfg[5] = mz;
    lago::g_gtile_cfg[6] = nthreads;
}
}
