// LDS-privatised 3D splat (adjoint of trilinear interpolation) -- gfx950.
//
// Same contract as interp_kernel_backward_3d of the reference
// (cuda/interp.cu:185-244 + atomicSplat include/interp.h:426-454), re-designed
// around the MI355X cost model: global float atomics execute at the memory side
// at ~1.3 TB/s of added bytes chip-wide (one 64-byte request per ~12 ns per CU),
// so issuing the reference's 8 atomics per voxel-channel caps the kernel at a
// fraction of the HBM roofline.  Here a workgroup owns a TX x TY x TZ tile of
// *source* voxels and accumulates their 8 corner contributions in an LDS window
// positioned at tile origin + displacement probed at the tile centre - margin.
// Smooth displacement fields keep
// nearly every corner inside the window; corners that fall outside take the
// global-atomic path, so any displacement is handled correctly.  The window is
// then flushed with one global atomic per *touched* cell, in wavefront rows of
// consecutive z whose start is 64-byte aligned (window z origin is a multiple of
// 16 cells).
//
// The LDS window accumulates in float64 even for float32 fields: on gfx950
// ds_add_f32 costs ~193 cycles per wave-instruction per CU while ds_add_f64 costs
// ~9 (measured, tools/probes/lds_atomic_rate.hip), so an fp32 window is LDS-atomic
// bound at ~0.7 ms for the 8 x 128^3 case.  Contributions are computed in the
// field precision exactly as the reference does, summed in double inside a tile,
// and rounded once at the flush.
//
// Each lane owns groups of VPL consecutive-z voxels of the tile (16-byte loads of u
// and grad_out, 16-byte stores of d_u).  The kernel is latency-bound (dependent HBM
// round trips per phase), so occupancy matters more than anything: this loop
// structure compiles to 99 VGPRs; a variant that kept the sample positions live
// across the barrier compiled to 217 and ran 1.7x slower.  VPL = 1 is the
// general-shape fallback.
//
// d_u (the analytic gradient term, include/interp.h:207-327) is produced by the
// same pass: the thread that owns a voxel owns its d_u entries, so the channel
// sum is a plain read-modify-write in ascending channel order, bit-identical to
// the reference's thread-owned accumulation.
#include "common.hpp"

#ifndef LAGO_NT_SPLAT_LD
#define LAGO_NT_SPLAT_LD 1   // grad_out and u of the C = 1 sheared splat are read once: -1.9 % (profiles/r04_cache_policy.md)
#endif
#ifndef LAGO_NT_SPLAT_MC_LD
#define LAGO_NT_SPLAT_MC_LD 1   // the same in the multi-channel form: interp_backward C = 3 -1.6 %
#endif
#ifndef LAGO_NT_SPLAT_ST
#define LAGO_NT_SPLAT_ST 1   // d_u stores non-temporal: -1.3 % (profiles/r04_cache_policy.md)
#endif

namespace lago {

// How a source voxel's sample position is obtained.
enum { POS_DISP_UNIT = 3,  // POS_DISP with dt == +-1 in float: positions are one float fma (common.hpp)
       POS_DISP = 0,     // x + dt*u(x): interp_backward (cuda/interp.cu:185-244)
       POS_AFFINE = 1,   // A(x - c) + T + c: affine_interp_backward's image splat (cuda/affine.cu:330-536)
       POS_REGRID = 2 }; // (X - C)S + O: regrid_backward (cuda/affine.cu:767-800)
struct PosArgs {
    const void *u, *A, *T;
    double dt, O[3], S[3];
    // how d_u starts (fused backward forms, include/lagomorph_hip.h lago_interp_backward_fused): 0 from zero (the
    // reference), 1 from the caller's d_u contents, 2 from addgo * grad_out[component] (needs nc == 3)
    int umode;
    double addgo;
    int gate;   // POS_AFFINE: 1 = only batch items whose matrix is NOT regular (common.hpp: affine_item_regular)
};

struct TileGeom {
    int nx, ny, nz;      // target grid (d_I, LDS window)
    int snx, sny, snz;   // source grid (grad_out, tiles); equals the target grid except for regrid
    int TX, TY, TZ;     // source tile (voxels)
    int WX, WY, WZ;     // LDS window (cells)
    int MX, MY, MZ;     // margin below the probed origin
    uint32_t ntx, nty, ntz, tiles_per_item, total;
    int rev;                           // launch direction (common.hpp)
    uint32_t tile_groups, win_cells;   // tile size in VPL-groups; window size in cells
                                       // atomics, bit2 no flush atomics; always 0 in production
    FastDiv d_tiles, d_tyz, d_tz;      // block id -> (n, bx, by, bz)
    FastDiv d_TyTzq, d_Tzq;            // tile group id -> (a, b, cq)
    FastDiv d_wey;                     // window row id -> (lx, ly)
};

// TX TY TZ(0 = auto) window margins MX MY MZ around the probed origin, threads per workgroup
static KnobArray<7> g_tile_cfg({0, 8, 0, 1, 1, 4, 512});

__device__ __forceinline__ void lds_add(double *p, double v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <typename R, int N>
struct alignas(sizeof(R) * N) SVec {
    R e[N];
};

template <typename R>
__device__ __forceinline__ R splat_half_extent(int n) {  // `.5*static_cast<Real>(n-1)`, cuda/affine.cu:42-43
    return (R)(.5 * (double)(R)(n - 1));
}

// Sample position of source voxel (i, j, k) for the analytic modes; arithmetic as in affine.hip.
template <typename R, int MODE>
__device__ __forceinline__ void analytic_pos(R &hx, R &hy, R &hz, int i, int j, int k, const R *An, const R *Tn,
                                             const PosArgs &pa, const TileGeom &tg) {
    if (MODE == POS_AFFINE) {
        const R ox = splat_half_extent<R>(tg.nx), oy = splat_half_extent<R>(tg.ny), oz = splat_half_extent<R>(tg.nz);
        const R fi = (R)i - ox, fj = (R)j - oy, fk = (R)k - oz;
        hx = lg_fma(An[2], fk, lg_fma(An[0], fi, An[1] * fj)) + Tn[0] + ox;
        hy = lg_fma(An[5], fk, lg_fma(An[3], fi, An[4] * fj)) + Tn[1] + oy;
        hz = lg_fma(An[8], fk, lg_fma(An[6], fi, An[7] * fj)) + Tn[2] + oz;
    } else {
        const R ox = splat_half_extent<R>(tg.snx), oy = splat_half_extent<R>(tg.sny), oz = splat_half_extent<R>(tg.snz);
        hx = lg_fma((R)i - ox, (R)pa.S[0], (R)pa.O[0]);
        hy = lg_fma((R)j - oy, (R)pa.S[1], (R)pa.O[1]);
        hz = lg_fma((R)k - oz, (R)pa.S[2], (R)pa.O[2]);
    }
}

// MC ("multi-channel"): displacement mode with d_u wanted and more than one channel, tile covered by
// ONE pass of the workgroup -- positions are computed once, d_u is accumulated in registers over the
// channels and written once (the channel-by-channel form re-reads u and read-modify-writes d_u per
// channel: 132 instead of 60 bytes per voxel at C = 3).  The per-channel expressions and their order
// are unchanged, so the result is bit-identical.
template <typename R, int MODE, bool BC, bool NEED_U, int NT, int VPL, bool MC = false>
__global__ __launch_bounds__(NT) void splat_tiled_kernel(R *__restrict__ d_I, R *__restrict__ d_u,
                                                         const R *__restrict__ go, const R *__restrict__ I,
                                                         PosArgs pa, int nc, TileGeom tg) {
    extern __shared__ __align__(16) unsigned char lago_smem[];
    double *win = reinterpret_cast<double *>(lago_smem);  // f64 accumulators: see the header note
    const int nx = tg.nx, ny = tg.ny, nz = tg.nz;
    const int snx = tg.snx, sny = tg.sny, snz = tg.snz;
    const size_t nv = (size_t)nx * ny * nz;       // target plane
    const size_t snv = (size_t)snx * sny * snz;   // source plane
    const double dt = pa.dt;
    constexpr bool DISP = MODE == POS_DISP || MODE == POS_DISP_UNIT;

    // workgroup -> (batch item, tile)
    const uint32_t L = block_order(blockIdx.x, tg.total, tg.rev);
    const uint32_t n = tg.d_tiles.div(L);
    uint32_t r = L - n * tg.tiles_per_item;
    const uint32_t bx = tg.d_tyz.div(r);
    r -= bx * (tg.nty * tg.ntz);
    const uint32_t by = tg.d_tz.div(r);
    const uint32_t bz = r - by * tg.ntz;
    const int x0 = bx * tg.TX, y0 = by * tg.TY, z0 = bz * tg.TZ;
    const int ex = min(tg.TX, snx - x0), ey = min(tg.TY, sny - y0), ez = min(tg.TZ, snz - z0);

    const R *un = DISP ? static_cast<const R *>(pa.u) + (size_t)n * 3 * snv : nullptr;
    const R *An = MODE == POS_AFFINE ? static_cast<const R *>(pa.A) + (size_t)n * 9 : nullptr;
    const R *Tn = MODE == POS_AFFINE ? static_cast<const R *>(pa.T) + (size_t)n * 3 : nullptr;
    if (MODE == POS_AFFINE && pa.gate) {   // (wave-uniform: the whole workgroup leaves)
        double Ai[9];
        if (affine_item_regular<R>(An, Ai)) return;
    }
    const R *In = BC ? I : I + (size_t)n * nc * nv;
    R *dIn = BC ? d_I : d_I + (size_t)n * nc * nv;
    const R *gon = go + (size_t)n * nc * snv;
    R *dun = NEED_U ? d_u + (size_t)n * 3 * nv : nullptr;

    // window origin (placement only affects speed, never the result): for a displacement field,
    // tile origin + displacement probed at the tile centre; for the analytic maps, the minimum over
    // the images of the tile's 8 corner voxels (exact for affine maps) -- minus a margin
    int bxo, byo, bzo;
    if (DISP) {
        const size_t sc = ((size_t)(x0 + ex / 2) * ny + (y0 + ey / 2)) * nz + (z0 + ez / 2);
        const float fdt = (float)dt;
        bxo = x0 + (int)floorf(fdt * (float)un[sc]);
        byo = y0 + (int)floorf(fdt * (float)un[sc + nv]);
        bzo = z0 + (int)floorf(fdt * (float)un[sc + 2 * nv]);
    } else {
        R mnx = (R)1e30, mny = (R)1e30, mnz = (R)1e30;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            R hx, hy, hz;
            analytic_pos<R, MODE>(hx, hy, hz, x0 + ((q & 4) ? ex - 1 : 0), y0 + ((q & 2) ? ey - 1 : 0),
                                  z0 + ((q & 1) ? ez - 1 : 0), An, Tn, pa, tg);
            mnx = hx < mnx ? hx : mnx;
            mny = hy < mny ? hy : mny;
            mnz = hz < mnz ? hz : mnz;
        }
        bxo = lg_floor(mnx); byo = lg_floor(mny); bzo = lg_floor(mnz);
    }
    const int wex = min(tg.WX, nx), wey = min(tg.WY, ny), wez = min(tg.WZ, nz);
    const int wx0 = max(0, min(bxo - tg.MX, nx - wex));
    const int wy0 = max(0, min(byo - tg.MY, ny - wey));
    const int wz0 = max(0, min((bzo - tg.MZ) & ~15, nz - wez));
    const int WY = tg.WY, WZ = tg.WZ;
    // floor corners whose whole footprint is unclamped and inside the window: lo <= f <= hi - 1
    const int ilox = wx0, iloy = wy0, iloz = wz0;   // the window lies inside the grid
    const int ispx = wex - 1, ispy = wey - 1, ispz = wez - 1;
    const int wbase = -((wx0 * WY + wy0) * WZ + wz0);

    // one voxel's contribution to the window / d_I (include/interp.h:431-453: floor corner,
    // sequentially flipped weights) -- shared by both loop structures below
    auto splat_voxel = [&](R hx, R hy, R hz, R diff, R *dIc) {
        const int fx = lg_floor(hx), fy = lg_floor(hy), fz = lg_floor(hz);
        const R dx = (R)1.f - (hx - (R)fx);
        const R dy = (R)1.f - (hy - (R)fy);
        const R dz = (R)1.f - (hz - (R)fz);
        R wgt[8];
        {
            R ddx = dx, ddy = dy, ddz = dz;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                wgt[q] = (ddx * ddy * ddz) * diff;
                ddz = (R)1.f - ddz;
                if (q & 1) ddy = (R)1.f - ddy;
                if ((q & 3) == 3) ddx = (R)1.f - ddx;
            }
        }
        const bool interior = (unsigned)(fx - ilox) < (unsigned)ispx && (unsigned)(fy - iloy) < (unsigned)ispy &&
                              (unsigned)(fz - iloz) < (unsigned)ispz;
        if (interior) {
            double *w0 = win + ((fx * WY + fy) * WZ + fz + wbase);
            double *w1 = w0 + WZ, *w2 = w0 + WY * WZ, *w3 = w2 + WZ;
            lds_add(w0, (double)wgt[0]);
            lds_add(w0 + 1, (double)wgt[1]);
            lds_add(w1, (double)wgt[2]);
            lds_add(w1 + 1, (double)wgt[3]);
            lds_add(w2, (double)wgt[4]);
            lds_add(w2 + 1, (double)wgt[5]);
            lds_add(w3, (double)wgt[6]);
            lds_add(w3 + 1, (double)wgt[7]);
        } else {
            const int gi[2] = {clamp1(fx, nx), clamp1(fx + 1, nx)};
            const int gj[2] = {clamp1(fy, ny), clamp1(fy + 1, ny)};
            const int gk[2] = {clamp1(fz, nz), clamp1(fz + 1, nz)};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int cx = gi[q >> 2], cy = gj[(q >> 1) & 1], cz = gk[q & 1];
                const int lx = cx - wx0, ly = cy - wy0, lz = cz - wz0;
                const bool inside = (unsigned)lx < (unsigned)wex && (unsigned)ly < (unsigned)wey &&
                                    (unsigned)lz < (unsigned)wez;
                if (inside) lds_add(&win[(lx * WY + ly) * WZ + lz], (double)wgt[q]);
                else atomic_add(dIc + ((size_t)cx * ny + cy) * nz + cz, wgt[q]);
            }
        }
    };
    // flush touched cells: one wave per window row (lx, ly) -- the row decode and both base
    // addresses are wave-uniform (scalar), the lanes run along z
    auto flush = [&](R *dIc) {
        const int lane = threadIdx.x & 63;
        const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const uint32_t nrows = (uint32_t)(wex * wey);
        for (uint32_t row = wv; row < nrows; row += NT / 64) {
            const uint32_t lx = tg.d_wey.div(row), ly = row - lx * (uint32_t)wey;
            const double *wrow = win + (lx * (uint32_t)WY + ly) * (uint32_t)WZ;
            R *grow = dIc + ((size_t)(wx0 + lx) * ny + (wy0 + ly)) * nz + wz0;
            for (int lz = lane; lz < wez; lz += 64) {
                const double acc = wrow[lz];
                if (acc != 0.0) atomic_add(grow + lz, (R)acc);
            }
        }
    };

    if constexpr (MC) {
        uint32_t sv[VPL];
        bool live[VPL];
        R hx[VPL], hy[VPL], hz[VPL], dux[VPL], duy[VPL], duz[VPL];
#pragma unroll
        for (int e = 0; e < VPL; ++e) {
            const uint32_t t = threadIdx.x + e * NT;
            const uint32_t a = tg.d_TyTzq.div(t);
            const uint32_t rr = t - a * (uint32_t)(tg.TY * tg.TZ);
            const uint32_t b = tg.d_Tzq.div(rr);
            const uint32_t cc = rr - b * (uint32_t)tg.TZ;
            live[e] = t < tg.tile_groups && (int)a < ex && (int)b < ey && (int)cc < ez;
            const int vi = x0 + a, vj = y0 + b, vk = z0 + cc;
            sv[e] = live[e] ? ((uint32_t)vi * sny + vj) * snz + vk : 0;
            hx[e] = sample_pos_t<R, MODE == POS_DISP_UNIT>(vi, dt, un[sv[e]]);
            hy[e] = sample_pos_t<R, MODE == POS_DISP_UNIT>(vj, dt, un[sv[e] + nv]);
            hz[e] = sample_pos_t<R, MODE == POS_DISP_UNIT>(vk, dt, un[sv[e] + 2 * nv]);
            dux[e] = duy[e] = duz[e] = (R)0;
            if (pa.umode == 1) {
                dux[e] = dun[sv[e]]; duy[e] = dun[sv[e] + nv]; duz[e] = dun[sv[e] + 2 * nv];
            } else if (pa.umode == 2) {
                const R ag = (R)pa.addgo;
                dux[e] = ag * gon[sv[e]]; duy[e] = ag * gon[snv + sv[e]]; duz[e] = ag * gon[2 * snv + sv[e]];
            }
        }
        Lerp3<R, false> Lq[VPL];  // gather geometry: once per voxel, reused by every channel (nz >= 2: host)
#pragma unroll
        for (int e = 0; e < VPL; ++e) Lq[e].setup(hx[e], hy[e], hz[e], nx, ny, nz);
        for (int c = 0; c < nc; ++c) {
            for (uint32_t f = threadIdx.x; f < tg.win_cells; f += NT) win[f] = 0.0;
            __syncthreads();
            const R *Ic = In + (size_t)c * nv;
            R *dIc = dIn + (size_t)c * nv;
            const R *gc = gon + (size_t)c * snv;
            R gv[VPL];
#pragma unroll
            for (int e = 0; e < VPL; ++e) gv[e] = gc[sv[e]];
#pragma unroll
            for (int e = 0; e < VPL; ++e) {
                if (!live[e]) continue;
                splat_voxel(hx[e], hy[e], hz[e], gv[e], dIc);
                R gx, gy, gz;
                Lq[e].grad(Ic, gx, gy, gz);
                const R diff = (R)((double)gv[e] * dt);  // cuda/interp.cu:230
                dux[e] = lg_fma(gx, diff, dux[e]);
                duy[e] = lg_fma(gy, diff, duy[e]);
                duz[e] = lg_fma(gz, diff, duz[e]);
            }
            __syncthreads();
            flush(dIc);
            __syncthreads();
        }
#pragma unroll
        for (int e = 0; e < VPL; ++e)
            if (live[e]) {
                dun[sv[e]] = dux[e];
                dun[sv[e] + nv] = duy[e];
                dun[sv[e] + 2 * nv] = duz[e];
            }
        return;
    }

    for (int c = 0; c < nc; ++c) {
        for (uint32_t f = threadIdx.x; f < tg.win_cells; f += NT) win[f] = 0.0;
        __syncthreads();
        const R *Ic = In + (size_t)c * nv;
        R *dIc = dIn + (size_t)c * nv;
        const R *gc = gon + (size_t)c * snv;
        // VPL voxels per thread per pass, slab-interleaved: voxel e of thread t is tile voxel
        // t + e*NT, so the lanes of a wave stay z-contiguous for every load, LDS atomic and gather
        // (lane-consecutive voxels per thread made the LDS atomics 4-way bank conflicted).
        for (uint32_t t0 = threadIdx.x; t0 < tg.tile_groups; t0 += NT * VPL) {
            size_t sv[VPL];
            int vi[VPL], vj[VPL], vk[VPL];
            bool live[VPL];
            R ux[VPL], uy[VPL], uz[VPL], gv[VPL], dux[VPL], duy[VPL], duz[VPL];
#pragma unroll
            for (int e = 0; e < VPL; ++e) {
                const uint32_t t = t0 + e * NT;
                const uint32_t a = tg.d_TyTzq.div(t);
                const uint32_t rr = t - a * (uint32_t)(tg.TY * tg.TZ);
                const uint32_t b = tg.d_Tzq.div(rr);
                const uint32_t cc = rr - b * (uint32_t)tg.TZ;
                live[e] = t < tg.tile_groups && (int)a < ex && (int)b < ey && (int)cc < ez;
                vi[e] = x0 + a; vj[e] = y0 + b; vk[e] = z0 + cc;
                sv[e] = live[e] ? ((size_t)vi[e] * sny + vj[e]) * snz + vk[e] : 0;
                if (DISP) {
                    ux[e] = un[sv[e]];
                    uy[e] = un[sv[e] + nv];
                    uz[e] = un[sv[e] + 2 * nv];
                }
                gv[e] = gc[sv[e]];
                if (NEED_U) {
                    if (c > 0 || pa.umode == 1) {
                        dux[e] = dun[sv[e]];
                        duy[e] = dun[sv[e] + nv];
                        duz[e] = dun[sv[e] + 2 * nv];
                    } else if (pa.umode == 2) {
                        const R ag = (R)pa.addgo;
                        dux[e] = ag * gon[sv[e]]; duy[e] = ag * gon[snv + sv[e]]; duz[e] = ag * gon[2 * snv + sv[e]];
                    } else {
                        dux[e] = duy[e] = duz[e] = (R)0;
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < VPL; ++e) {
                if (!live[e]) continue;
                R hx, hy, hz;
                if (DISP) {
                    hx = sample_pos_t<R, MODE == POS_DISP_UNIT>(vi[e], dt, ux[e]);
                    hy = sample_pos_t<R, MODE == POS_DISP_UNIT>(vj[e], dt, uy[e]);
                    hz = sample_pos_t<R, MODE == POS_DISP_UNIT>(vk[e], dt, uz[e]);
                } else {
                    analytic_pos<R, MODE>(hx, hy, hz, vi[e], vj[e], vk[e], An, Tn, pa, tg);
                }
                R diff = gv[e];
                splat_voxel(hx, hy, hz, diff, dIc);
                if (NEED_U) {
                    Lerp3<R, false> Lq;  // nz >= 2 guaranteed by the host: lets the compiler batch the gathers
                    Lq.setup(hx, hy, hz, nx, ny, nz);
                    R gx, gy, gz;
                    Lq.grad(Ic, gx, gy, gz);
                    diff = (R)((double)diff * dt);  // cuda/interp.cu:230
                    dun[sv[e]] = lg_fma(gx, diff, dux[e]);
                    dun[sv[e] + nv] = lg_fma(gy, diff, duy[e]);
                    dun[sv[e] + 2 * nv] = lg_fma(gz, diff, duz[e]);
                }
            }
        }
        __syncthreads();
        flush(dIc);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Sheared-window splat for float32 displacement fields (interp_backward, 3D) -- the hot form.
//
// What the experiments of round 2 showed (profiles/r02_splat_experiments.md, tools/probes/splat_floor.hip,
// splat_stages.hip): the bare scheme "tile -> float64 LDS window -> atomic flush" runs the 8 x 128^3 case in
// 56 us at 1024 threads per workgroup, 73 us at 512; positions + weights bring it to 85 us; the kernel above
// needs 230 us because (1) a window placed once per 4 x 8 x nz tile loses the footprints of ~10 % of the voxels
// of a smooth field -- the displacement drifts by +-2 voxels along a 128-voxel row, and every row end needs a
// clamp -- and each lost corner becomes a lone global atomic: they cost 2-3 times the atomic requests of the whole
// flush; (2) its 4-voxels-per-lane unrolling needs ~90 VGPRs, which caps a CU at 16 waves.
// Hence here:
//  * the window is SHEARED along z: every 16-cell z segment of the window has its own (x, y) origin, the
//    displacement probed at the tile's centre column at that height.  A margin of one cell then holds all but
//    ~0.3 % of the footprints of a smooth field (10 % with one origin per tile);
//  * the window may reach one cell beyond the grid in x and y and the flush folds those cells onto the border
//    (what the reference's clamp does), z cells are clamped when they are added: a row end or a face of the
//    volume is no special case;
//  * one voxel per lane per pass of a rolled loop, 1024 threads, under 64 VGPRs: 32 waves per CU;
//  * what still misses the window takes the reference's clamped global atomics, corner by corner.
// Arithmetic (positions, sequentially flipped weights, gradient expressions) is that of the kernel above, so d_u
// is bit-identical and d_I differs only through the order of the float64 window adds.
struct ShearGeom {
    int nx, ny, nz;
    int TX, TY, TZ;        // source tile
    int WX, WY, WZ;        // window cells; x / y in virtual coordinates [-1, n], z inside the grid
    int MX, MY, MZ;
    int nseg;              // 16-cell z segments of the window
    uint32_t ntx, nty, ntz, tiles_per_item, total, tile_vox, win_cells;
    int rev;               // launch direction (common.hpp)
    FastDiv d_tiles, d_tyz, d_tz, d_TyTz, d_Tz, d_wy;
};

template <bool UNIT>
__device__ __forceinline__ float shear_pos(int base, double dt, float u) {
    if (UNIT) return __builtin_fmaf((float)dt, u, (float)base);  // one rounding of the exact sum (common.hpp: sample_pos_t)
    return (float)__builtin_fma(dt, (double)u, (double)base);
}

// (<= 80 SGPRs and <= 64 VGPRs: two 1024-thread workgroups per CU, 32 waves; at 81+ SGPRs the CU admits only one)
// VPL = 0: rolled loop over the tile's voxels, d_u read-modify-written per channel.  VPL > 0 (several channels with
// d_u wanted, tile covered by VPL passes of the workgroup): the passes are unrolled and each voxel's d_u stays in
// registers over the channel loop -- 12 instead of 12 + 24 (C - 1) bytes per voxel of d_u traffic.
template <int NT, bool NEED_U, bool UNIT, bool BC, int VPL = 0>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_num_sgpr(80))) void splat_shear_kernel(float *__restrict__ d_I, float *__restrict__ d_u,
                                                         const float *__restrict__ go, const float *__restrict__ I,
                                                         const float *__restrict__ u, double dt, int nc, ShearGeom sg,
                                                         int umode, float addgo) {
    extern __shared__ __align__(16) unsigned char lago_smem[];
    double *win = reinterpret_cast<double *>(lago_smem);
    int2 *org = reinterpret_cast<int2 *>(lago_smem + (size_t)sg.win_cells * 8);  // (x, y) origin per z segment
    const int nx = sg.nx, ny = sg.ny, nz = sg.nz;
    const uint32_t nv = (uint32_t)nx * ny * nz;
    const uint32_t planeB = nv * 4u;

    // workgroup -> (batch item, tile)
    const uint32_t L = block_order(blockIdx.x, sg.total, sg.rev);
    const uint32_t n = sg.d_tiles.div(L);
    uint32_t r = L - n * sg.tiles_per_item;
    const uint32_t bx = sg.d_tyz.div(r);
    r -= bx * (sg.nty * sg.ntz);
    const uint32_t by = sg.d_tz.div(r);
    const uint32_t bz = r - by * sg.ntz;
    const int x0 = bx * sg.TX, y0 = by * sg.TY, z0 = bz * sg.TZ;
    const int ex = min(sg.TX, nx - x0), ey = min(sg.TY, ny - y0), ez = min(sg.TZ, nz - z0);

    const float *un = u + (size_t)n * 3 * nv;
    const float *In = BC ? I : I + (size_t)n * nc * nv;
    float *dIn = BC ? d_I : d_I + (size_t)n * nc * nv;
    const float *gon = go + (size_t)n * nc * nv;
    float *dun = NEED_U ? d_u + (size_t)n * 3 * nv : nullptr;

    const int WX = sg.WX, WY = sg.WY, WZ = sg.WZ;
    const int wez = min(WZ, nz);
    // window z range: 16-aligned start (flush rows start on 64-byte boundaries), inside the grid
    const int cxs = x0 + ex / 2, cys = y0 + ey / 2;
    int wz0;
    {
        const float fdt = (float)dt;
        const size_t sc = ((size_t)cxs * ny + cys) * nz + (z0 + ez / 2);
        const int bzo = z0 + (int)floorf(fdt * un[sc + 2 * (size_t)nv]);
        wz0 = max(0, min((bzo - sg.MZ) & ~15, nz - wez));
    }
    // origin of every z segment: the displacement of the tile's centre column at the segment's height
    if ((int)threadIdx.x < sg.nseg) {
        const float fdt = (float)dt;
        const int zc = min(wz0 + (int)threadIdx.x * 16 + 8, nz - 1);
        const size_t sc = ((size_t)cxs * ny + cys) * nz + zc;
        int2 o;
        o.x = max(-1, min(x0 + (int)floorf(fdt * un[sc]) - sg.MX, nx + 1 - WX));
        o.y = max(-1, min(y0 + (int)floorf(fdt * un[sc + nv]) - sg.MY, ny + 1 - WY));
        org[threadIdx.x] = o;
    }
    const uint32_t sxB = (uint32_t)(WY * WZ) * 8u, syB = (uint32_t)WZ * 8u;            // window strides in bytes
    const uint32_t gxB = (uint32_t)ny * nz * 4u, gyB = (uint32_t)nz * 4u;              // grid strides in bytes
    const uint32_t wxu1 = (uint32_t)(WX - 1), wyu1 = (uint32_t)(WY - 1), wezu = (uint32_t)wez;

    constexpr int NV = VPL > 0 ? VPL : 1;
    float rux[NV], ruy[NV], ruz[NV];   // VPL > 0: the d_u sums of this thread's voxels
    for (int c = 0; c < nc; ++c) {
        for (uint32_t f = threadIdx.x; f < sg.win_cells; f += NT) win[f] = 0.0;
        __syncthreads();
        const float *Ic = In + (size_t)c * nv;
        const float *gc = gon + (size_t)c * nv;
        const BufRsrc rdI = make_rsrc(dIn + (size_t)c * nv, planeB);
        const BufRsrc rI = make_rsrc(Ic, planeB);
#pragma unroll
        for (int it = 0; it < (VPL > 0 ? VPL : (int)((sg.tile_vox + NT - 1) / NT)); ++it) {
            uint32_t tt = threadIdx.x + (uint32_t)it * NT;
            if (VPL > 0) {
                // the unrolled passes stay one after the other and recompute their geometry per channel: hoisting it
                // out of the channel loop (it does not depend on c) or every pass's loads to the top costs 128 VGPRs
                // and spills -- and with them the second workgroup of the CU
                asm volatile("" : "+v"(tt)::"memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            if (tt >= sg.tile_vox) break;
            const uint32_t a = sg.d_TyTz.div(tt);
            const uint32_t rr = tt - a * (uint32_t)(sg.TY * sg.TZ);
            const uint32_t b = sg.d_Tz.div(rr);
            const uint32_t kk = rr - b * (uint32_t)sg.TZ;
            if ((int)a >= ex || (int)b >= ey || (int)kk >= ez) continue;
            const int vi = x0 + a, vj = y0 + b, vk = z0 + kk;
            const uint32_t sv = ((uint32_t)vi * ny + vj) * nz + vk;
            const float gv = ld_pol<LAGO_NT_SPLAT_LD>(gc + sv);
            const float hx = shear_pos<UNIT>(vi, dt, ld_pol<LAGO_NT_SPLAT_LD>(un + sv));
            const float hy = shear_pos<UNIT>(vj, dt, ld_pol<LAGO_NT_SPLAT_LD>(un + sv + nv));
            const float hz = shear_pos<UNIT>(vk, dt, ld_pol<LAGO_NT_SPLAT_LD>(un + sv + 2 * (size_t)nv));
            const int fx = lg_floor(hx), fy = lg_floor(hy), fz = lg_floor(hz);
            const float t = hx - (float)fx, uu = hy - (float)fy, v = hz - (float)fz;
            const float omt = 1.f - t, omu = 1.f - uu, omv = 1.f - v;
            // sequentially flipped weights (include/interp.h:431-453): x outer, y, z inner
            float wq[8];
            {
                float ddx = omt, ddy = omu, ddz = omv;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    wq[q] = (ddx * ddy * ddz) * gv;
                    ddz = 1.f - ddz;
                    if (q & 1) ddy = 1.f - ddy;
                    if ((q & 3) == 3) ddx = 1.f - ddx;
                }
            }
            // the two z cells, clamped as the reference clamps them, and the window segment each falls in
            const int cz0 = clamp1(fz, nz), cz1 = clamp1(fz + 1, nz);
            const uint32_t lz0 = (uint32_t)(cz0 - wz0), lz1 = (uint32_t)(cz1 - wz0);
            bool inwin = lz0 < wezu && lz1 < wezu;
            int2 o0 = {0, 0}, o1 = {0, 0};
            if (inwin) {
                o0 = org[lz0 >> 4];
                o1 = org[lz1 >> 4];
            }
            const uint32_t lx0 = (uint32_t)(fx - o0.x), ly0 = (uint32_t)(fy - o0.y);
            const uint32_t lx1 = (uint32_t)(fx - o1.x), ly1 = (uint32_t)(fy - o1.y);
            inwin = inwin && lx0 < wxu1 && ly0 < wyu1 && lx1 < wxu1 && ly1 < wyu1;
            if (inwin) {
                const uint32_t a0 = __umul24(lx0, sxB) + __umul24(ly0, syB) + lz0 * 8u;
                const uint32_t a1 = __umul24(lx1, sxB) + __umul24(ly1, syB) + lz1 * 8u;
                lds_add(reinterpret_cast<double *>(lago_smem + a0), (double)wq[0]);
                lds_add(reinterpret_cast<double *>(lago_smem + a1), (double)wq[1]);
                lds_add(reinterpret_cast<double *>(lago_smem + a0 + syB), (double)wq[2]);
                lds_add(reinterpret_cast<double *>(lago_smem + a1 + syB), (double)wq[3]);
                lds_add(reinterpret_cast<double *>(lago_smem + a0 + sxB), (double)wq[4]);
                lds_add(reinterpret_cast<double *>(lago_smem + a1 + sxB), (double)wq[5]);
                lds_add(reinterpret_cast<double *>(lago_smem + a0 + sxB + syB), (double)wq[6]);
                lds_add(reinterpret_cast<double *>(lago_smem + a1 + sxB + syB), (double)wq[7]);
            } else {
                // beyond the window: the reference's clamped global atomics (include/interp.h:330-401, :431-453)
                const uint32_t X0 = __umul24((uint32_t)clamp1(fx, nx), gxB), X1 = __umul24((uint32_t)clamp1(fx + 1, nx), gxB);
                const uint32_t Y0 = __umul24((uint32_t)clamp1(fy, ny), gyB), Y1 = __umul24((uint32_t)clamp1(fy + 1, ny), gyB);
                const uint32_t Z0 = (uint32_t)cz0 * 4u, Z1 = (uint32_t)cz1 * 4u;
                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[0], rdI, X0 + Y0 + Z0, 0, 0);
                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[1], rdI, X0 + Y0 + Z1, 0, 0);
                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[2], rdI, X0 + Y1 + Z0, 0, 0);
                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[3], rdI, X0 + Y1 + Z1, 0, 0);
                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[4], rdI, X1 + Y0 + Z0, 0, 0);
                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[5], rdI, X1 + Y0 + Z1, 0, 0);
                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[6], rdI, X1 + Y1 + Z0, 0, 0);
                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[7], rdI, X1 + Y1 + Z1, 0, 0);
            }
            if (NEED_U) {
                float gx, gy, gz;
                {
                    // rows (fx,fy) (fx+1,fy) (fx+1,fy+1) (fx,fy+1), each clamped by itself as Lerp3::setup does (common.hpp) -- a
                    // sample at a face of the volume is no special case (a lane that recomputed its position there stalled its
                    // wave on three exposed memory round trips, profiles/r06_march_window.md); the z pair is fetched at
                    // zb = clamp(fz, 0, nz-2) and picked as Lerp3 does: beyond the upper border both corners are the pair's
                    // high half, below the lower border both are its low half
                    const int cx0 = clamp1(fx, nx), cy0 = clamp1(fy, ny);
                    const uint32_t dX = clamp1(fx + 1, nx) != cx0 ? gxB : 0u, dY = clamp1(fy + 1, ny) != cy0 ? gyB : 0u;
                    const int zb = lg_med3(fz, 0, nz - 2);
                    const bool f_hi = fz > nz - 2, c_lo = fz < 0;
                    const uint32_t o = __umul24((uint32_t)cx0, gxB) + __umul24((uint32_t)cy0, gyB) + (uint32_t)zb * 4u;
                    float l0, l1, l2, l3, h0, h1, h2, h3;
                    buf_load2s(rI, o, 0u, l0, h0);
                    buf_load2s(rI, o + dX, 0u, l1, h1);
                    buf_load2s(rI, o + dX + dY, 0u, l2, h2);
                    buf_load2s(rI, o + dY, 0u, l3, h3);
                    const float c0 = f_hi ? h0 : l0, c1 = f_hi ? h1 : l1, c2 = f_hi ? h2 : l2, c3 = f_hi ? h3 : l3;
                    const float c4 = c_lo ? l0 : h0, c5 = c_lo ? l1 : h1, c6 = c_lo ? l2 : h2, c7 = c_lo ? l3 : h3;
                    // include/interp.h:315-326
                    gx = lg_fma(omv, lg_fma(omu, c1 - c0, uu * (c2 - c3)), v * lg_fma(omu, c5 - c4, uu * (c6 - c7)));
                    gy = lg_fma(omv, lg_fma(omt, c3 - c0, t * (c2 - c1)), v * lg_fma(omt, c7 - c4, t * (c6 - c5)));
                    gz = lg_fma(omu, lg_fma(omt, c4 - c0, t * (c5 - c1)), uu * lg_fma(omt, c7 - c3, t * (c6 - c2)));
                }
                // cuda/interp.cu:230: (Real)((double)diff * dt); for dt = +-1 that is +-diff exactly
                const float diff = UNIT ? (float)dt * gv : (float)((double)gv * dt);
                // ascending channel order, as the reference's thread-owned accumulation; the start value is zero
                // for the reference operator, the caller's d_u or addgo * grad_out for the fused backward forms
                float ix = 0.f, iy = 0.f, iz = 0.f;
                if (VPL > 0 && c > 0) {
                    ix = rux[it % NV]; iy = ruy[it % NV]; iz = ruz[it % NV];
                } else if (c > 0 || umode == 1) {
                    ix = dun[sv]; iy = dun[sv + nv]; iz = dun[sv + 2 * (size_t)nv];
                } else if (umode == 2) {
                    ix = addgo * gon[sv]; iy = addgo * gon[sv + nv]; iz = addgo * gon[sv + 2 * (size_t)nv];
                }
                ix = lg_fma(gx, diff, ix);
                iy = lg_fma(gy, diff, iy);
                iz = lg_fma(gz, diff, iz);
                if (VPL > 0 && c + 1 < nc) {
                    rux[it % NV] = ix; ruy[it % NV] = iy; ruz[it % NV] = iz;
                } else {
                    st_pol<LAGO_NT_SPLAT_ST>(&dun[sv], ix); st_pol<LAGO_NT_SPLAT_ST>(&dun[sv + nv], iy); st_pol<LAGO_NT_SPLAT_ST>(&dun[sv + 2 * (size_t)nv], iz);
                }
            }
        }
        __syncthreads();
        // flush touched cells: one wave per window row (lx, ly), lanes along z; the (x, y) a cell belongs to is
        // its segment's origin + (lx, ly), folded onto the grid (the clamp of the reference)
        {
            const int lane = threadIdx.x & 63;
            const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
            const uint32_t nrows = (uint32_t)(WX * WY);
            for (uint32_t row = wave; row < nrows; row += NT / 64) {
                const uint32_t lx = sg.d_wy.div(row), ly = row - lx * (uint32_t)WY;
                const double *wrow = win + row * (uint32_t)WZ;
                for (int lz = lane; lz < wez; lz += 64) {
                    const double acc = wrow[lz];
                    if (acc != 0.0) {
                        const int2 o = org[lz >> 4];
                        const uint32_t off = __umul24((uint32_t)clamp1(o.x + (int)lx, nx), gxB) +
                                             __umul24((uint32_t)clamp1(o.y + (int)ly, ny), gyB) + (uint32_t)(wz0 + lz) * 4u;
                        (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32((float)acc, rdI, off, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Geometry-once form of the sheared-window splat for SEVERAL channels with d_u wanted (the reverse sweep of expmap
// splats three-channel fields: 33 % of the 160^3 atlas step, profiles/r02_atlas160_kernel_stats.md).
// splat_shear_kernel<..., VPL> recomputes a voxel's whole geometry per channel -- u re-read (1.54 x the algorithmic
// HBM traffic), the float64 position of the non-unit step, floors, window addressing with two origin look-ups, the
// gather offset: about 150 of its 245 vector instructions per voxel-channel.  Here the tile is covered by at most
// VPL = 2 passes of the 1024-thread workgroup and each lane keeps, per voxel, across the channel loop:
//   the two LDS byte addresses of its footprint's z cells (or NOWIN), the gather byte offset with the two z-border
//   flags in its low bits (rows clamped one by one: no recomputing lane), the three fractions, the voxel index and the three d_u sums
// -- 10 registers per voxel.  Per channel what is left is: one grad_out load, the eight sequentially flipped weights,
// eight float64 LDS adds, four pair gathers, the gradient expressions.  Voxels whose footprint leaves the window or
// whose rows are clamped (well under 1 % of a smooth field) recompute their position from u per channel and take the
// reference's clamped paths, as in splat_shear_kernel.  The flush re-zeroes the cells it reads, so a channel costs
// two barriers instead of three.  Arithmetic per voxel and channel is that of splat_shear_kernel: d_u bit-identical.
template <int NT, bool UNIT, bool BC, int VPL>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8, 8))) void splat_shear_mc_kernel(
    float *__restrict__ d_I, float *__restrict__ d_u, const float *__restrict__ go, const float *__restrict__ I,
    const float *__restrict__ u, double dt, int nc, ShearGeom sg, int umode, float addgo) {
    extern __shared__ __align__(16) unsigned char lago_smem[];
    double *win = reinterpret_cast<double *>(lago_smem);
    int2 *org = reinterpret_cast<int2 *>(lago_smem + (size_t)sg.win_cells * 8);  // (x, y) origin per z segment
    const int nx = sg.nx, ny = sg.ny, nz = sg.nz;
    const uint32_t nv = (uint32_t)nx * ny * nz;
    const uint32_t planeB = nv * 4u;
    constexpr uint32_t NOWIN = 0xffffffffu, NOROW = 0xffffffffu, DEAD = 0xffffffffu;

    // workgroup -> (batch item, tile)
    const uint32_t L = block_order(blockIdx.x, sg.total, sg.rev);
    const uint32_t n = sg.d_tiles.div(L);
    uint32_t r = L - n * sg.tiles_per_item;
    const uint32_t bx = sg.d_tyz.div(r);
    r -= bx * (sg.nty * sg.ntz);
    const uint32_t by = sg.d_tz.div(r);
    const uint32_t bz = r - by * sg.ntz;
    const int x0 = bx * sg.TX, y0 = by * sg.TY, z0 = bz * sg.TZ;
    const int ex = min(sg.TX, nx - x0), ey = min(sg.TY, ny - y0), ez = min(sg.TZ, nz - z0);

    const float *un = u + (size_t)n * 3 * nv;
    const float *In = BC ? I : I + (size_t)n * nc * nv;
    float *dIn = BC ? d_I : d_I + (size_t)n * nc * nv;
    const float *gon = go + (size_t)n * nc * nv;
    float *dun = d_u + (size_t)n * 3 * nv;

    const int WX = sg.WX, WY = sg.WY, WZ = sg.WZ;
    const int wez = min(WZ, nz);
    const int cxs = x0 + ex / 2, cys = y0 + ey / 2;
    int wz0;
    {
        const float fdt = (float)dt;
        const size_t sc = ((size_t)cxs * ny + cys) * nz + (z0 + ez / 2);
        const int bzo = z0 + (int)floorf(fdt * un[sc + 2 * (size_t)nv]);
        wz0 = max(0, min((bzo - sg.MZ) & ~15, nz - wez));
    }
    if ((int)threadIdx.x < sg.nseg) {
        const float fdt = (float)dt;
        const int zc = min(wz0 + (int)threadIdx.x * 16 + 8, nz - 1);
        const size_t sc = ((size_t)cxs * ny + cys) * nz + zc;
        int2 o;
        o.x = max(-1, min(x0 + (int)floorf(fdt * un[sc]) - sg.MX, nx + 1 - WX));
        o.y = max(-1, min(y0 + (int)floorf(fdt * un[sc + nv]) - sg.MY, ny + 1 - WY));
        org[threadIdx.x] = o;
    }
    for (uint32_t f = threadIdx.x; f < sg.win_cells; f += NT) win[f] = 0.0;
    __syncthreads();
    const uint32_t sxB = (uint32_t)(WY * WZ) * 8u, syB = (uint32_t)WZ * 8u;            // window strides in bytes
    const uint32_t gxB = (uint32_t)ny * nz * 4u, gyB = (uint32_t)nz * 4u;              // grid strides in bytes
    const uint32_t wxu1 = (uint32_t)(WX - 1), wyu1 = (uint32_t)(WY - 1), wezu = (uint32_t)wez;

    // ---- per-voxel geometry, once
    uint32_t SV[VPL], A0[VPL], A1[VPL], OG[VPL];
    float FT[VPL], FU[VPL], FV[VPL], rux[VPL], ruy[VPL], ruz[VPL];
#pragma unroll
    for (int it = 0; it < VPL; ++it) {
        const uint32_t tt = threadIdx.x + (uint32_t)it * NT;
        const uint32_t a = sg.d_TyTz.div(tt);
        const uint32_t rr = tt - a * (uint32_t)(sg.TY * sg.TZ);
        const uint32_t b = sg.d_Tz.div(rr);
        const uint32_t kk = rr - b * (uint32_t)sg.TZ;
        SV[it] = DEAD;
        A0[it] = A1[it] = NOWIN;
        OG[it] = NOROW;
        FT[it] = FU[it] = FV[it] = rux[it] = ruy[it] = ruz[it] = 0.f;
        if (tt >= sg.tile_vox || (int)a >= ex || (int)b >= ey || (int)kk >= ez) continue;
        const int vi = x0 + a, vj = y0 + b, vk = z0 + kk;
        const uint32_t sv = ((uint32_t)vi * ny + vj) * nz + vk;
        SV[it] = sv;
        const float hx = shear_pos<UNIT>(vi, dt, ld_pol<LAGO_NT_SPLAT_MC_LD>(un + sv));
        const float hy = shear_pos<UNIT>(vj, dt, ld_pol<LAGO_NT_SPLAT_MC_LD>(un + sv + nv));
        const float hz = shear_pos<UNIT>(vk, dt, ld_pol<LAGO_NT_SPLAT_MC_LD>(un + sv + 2 * (size_t)nv));
        const int fx = lg_floor(hx), fy = lg_floor(hy), fz = lg_floor(hz);
        FT[it] = hx - (float)fx;
        FU[it] = hy - (float)fy;
        FV[it] = hz - (float)fz;
        // the two z cells, clamped as the reference clamps them, and the window segment each falls in
        const int cz0 = clamp1(fz, nz), cz1 = clamp1(fz + 1, nz);
        const uint32_t lz0 = (uint32_t)(cz0 - wz0), lz1 = (uint32_t)(cz1 - wz0);
        bool inwin = lz0 < wezu && lz1 < wezu;
        int2 o0 = {0, 0}, o1 = {0, 0};
        if (inwin) {
            o0 = org[lz0 >> 4];
            o1 = org[lz1 >> 4];
        }
        const uint32_t lx0 = (uint32_t)(fx - o0.x), ly0 = (uint32_t)(fy - o0.y);
        const uint32_t lx1 = (uint32_t)(fx - o1.x), ly1 = (uint32_t)(fy - o1.y);
        inwin = inwin && lx0 < wxu1 && ly0 < wyu1 && lx1 < wxu1 && ly1 < wyu1;
        if (inwin) {
            A0[it] = __umul24(lx0, sxB) + __umul24(ly0, syB) + lz0 * 8u;
            A1[it] = __umul24(lx1, sxB) + __umul24(ly1, syB) + lz1 * 8u;
        }
        {
            // the four (x, y) rows of the footprint, each clamped by itself as Lerp3::setup does (common.hpp): base row
            // (clamp fx, clamp fy) at zb = clamp(fz, 0, nz-2) with the two z-border flags in its low bits; whether the
            // x + 1 / y + 1 rows coincide with it (a sample at or beyond a face) rides in the two top bits of A1
            const int cx0 = clamp1(fx, nx), cy0 = clamp1(fy, ny);
            const int zb = lg_med3(fz, 0, nz - 2);
            const uint32_t f_hi = fz > nz - 2 ? 1u : 0u, c_lo = fz < 0 ? 2u : 0u;
            OG[it] = (__umul24((uint32_t)cx0, gxB) + __umul24((uint32_t)cy0, gyB) + (uint32_t)zb * 4u) | f_hi | c_lo;
            A1[it] = (A1[it] & 0x3fffffffu) | (clamp1(fx + 1, nx) == cx0 ? 0x80000000u : 0u) | (clamp1(fy + 1, ny) == cy0 ? 0x40000000u : 0u);
        }
        if (umode == 1) {
            rux[it] = dun[sv]; ruy[it] = dun[sv + nv]; ruz[it] = dun[sv + 2 * (size_t)nv];
        } else if (umode == 2) {
            rux[it] = addgo * gon[sv]; ruy[it] = addgo * gon[sv + nv]; ruz[it] = addgo * gon[sv + 2 * (size_t)nv];
        }
    }

    // ---- channels.  What bounds these kernels is the number of DEPENDENT memory round trips inside a barrier-phased
    // workgroup (two workgroups per CU), not instructions, LDS atomics or HBM bytes (profiles/r03_splat_latency.md):
    //  * the next channel's grad_out values are requested before the barrier / flush / barrier of the current one;
    //  * a channel's corner-pair gathers are all issued first (unconditional buffer loads: an out-of-range offset --
    //    DEAD / NOROW are 0xffffffff -- returns 0; a load inside a branch would be waited for at the branch's join),
    //    so that the weights and the LDS adds of every pass run under their latency;
    //  * a footprint that leaves the window takes its clamped cells from the kept gather offset (no reload of u); only
    //    samples whose rows are clamped -- positions outside the grid -- recompute their position.
    float pgv[VPL];
    auto request_gv = [&](int c) {
        const BufRsrc rg = make_rsrc(gon + (size_t)c * nv, planeB);
#pragma unroll
        for (int it = 0; it < VPL; ++it)
            pgv[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, SV[it] == DEAD ? DEAD : SV[it] * 4u, 0, LAGO_NT_SPLAT_MC_LD ? 2 : 0));
    };
    request_gv(0);
    for (int c = 0; c < nc; ++c) {
        const float *Ic = In + (size_t)c * nv;
        const BufRsrc rdI = make_rsrc(dIn + (size_t)c * nv, planeB);
        const BufRsrc rI = make_rsrc(Ic, planeB);
        float gvs[VPL];
#pragma unroll
        for (int it = 0; it < VPL; ++it) gvs[it] = pgv[it];
        if (c + 1 < nc) request_gv(c + 1);
#pragma unroll
        for (int it = 0; it < VPL; ++it) {
            if (VPL > 1 && it) __builtin_amdgcn_sched_barrier(0);  // one pass after the other (register pressure)
            // the kept geometry is made opaque once per channel: otherwise everything derived from it (eight window
            // addresses, four gather offsets, 1 - t ..., per voxel) is hoisted out of the channel loop into registers
            // (111 VGPRs instead of 64: one workgroup per CU)
            asm volatile("" : "+v"(A0[it]), "+v"(A1[it]), "+v"(OG[it]), "+v"(FT[it]), "+v"(FU[it]), "+v"(FV[it]), "+v"(SV[it]));
            unsigned long long pq[4];   // the four corner pairs: rows (fx,fy) (fx+1,fy) (fx+1,fy+1) (fx,fy+1)
            const uint32_t dXv = (A1[it] & 0x80000000u) ? 0u : gxB, dYv = (A1[it] & 0x40000000u) ? 0u : gyB;
            {
                const uint32_t o = OG[it] & ~3u;   // (a dead lane's 0xfffffffc reads 0 or a wrapped in-range cell: unused)
                pq[0] = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(rI, o, 0u, 0));
                pq[1] = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(rI, o + dXv, 0u, 0));
                pq[2] = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(rI, o + dXv + dYv, 0u, 0));
                pq[3] = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(rI, o + dYv, 0u, 0));
            }
            const uint32_t sv = SV[it];
            const float gv = gvs[it];
            if (sv == DEAD) continue;
            const float omt = 1.f - FT[it], omu = 1.f - FU[it], omv = 1.f - FV[it];
            // sequentially flipped weights (include/interp.h:431-453): x outer, y, z inner
            float wq[8];
            {
                float ddx = omt, ddy = omu, ddz = omv;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    wq[q] = (ddx * ddy * ddz) * gv;
                    ddz = 1.f - ddz;
                    if (q & 1) ddy = 1.f - ddy;
                    if ((q & 3) == 3) ddx = 1.f - ddx;
                }
            }
            const uint32_t a0 = A0[it], a1 = A1[it] & 0x3fffffffu;
            if (a0 != NOWIN) {
                lds_add(reinterpret_cast<double *>(lago_smem + a0), (double)wq[0]);
                lds_add(reinterpret_cast<double *>(lago_smem + a1), (double)wq[1]);
                lds_add(reinterpret_cast<double *>(lago_smem + a0 + syB), (double)wq[2]);
                lds_add(reinterpret_cast<double *>(lago_smem + a1 + syB), (double)wq[3]);
                lds_add(reinterpret_cast<double *>(lago_smem + a0 + sxB), (double)wq[4]);
                lds_add(reinterpret_cast<double *>(lago_smem + a1 + sxB), (double)wq[5]);
                lds_add(reinterpret_cast<double *>(lago_smem + a0 + sxB + syB), (double)wq[6]);
                lds_add(reinterpret_cast<double *>(lago_smem + a1 + sxB + syB), (double)wq[7]);
            } else {
                // beyond the window: the reference's clamped global atomics (include/interp.h:330-401, :431-453)
                uint32_t X0, X1, Y0, Y1, Z0, Z1;
                {
                    // the clamped cells follow from the gather offset (clamped fx, fy; zb), the row flags and the z-border flags
                    const uint32_t o = OG[it] & ~3u;
                    X0 = o; X1 = o + dXv; Y0 = 0u; Y1 = dYv;
                    Z0 = (OG[it] & 1u) ? 4u : 0u;              // floor beyond the upper face: both cells are nz - 1 = zb + 1
                    Z1 = (OG[it] & 2u) ? 0u : 4u;              // floor below the lower face: both cells are 0 = zb
                }
                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[0], rdI, X0 + Y0 + Z0, 0, 0);
                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[1], rdI, X0 + Y0 + Z1, 0, 0);
                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[2], rdI, X0 + Y1 + Z0, 0, 0);
                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[3], rdI, X0 + Y1 + Z1, 0, 0);
                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[4], rdI, X1 + Y0 + Z0, 0, 0);
                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[5], rdI, X1 + Y0 + Z1, 0, 0);
                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[6], rdI, X1 + Y1 + Z0, 0, 0);
                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[7], rdI, X1 + Y1 + Z1, 0, 0);
            }
            const float t = FT[it], uu = FU[it], v = FV[it];
            float gx, gy, gz;
            {
                const bool f_hi = OG[it] & 1u, c_lo = OG[it] & 2u;
                auto lo = [](unsigned long long x) { return __builtin_bit_cast(float, (unsigned int)x); };
                auto hi = [](unsigned long long x) { return __builtin_bit_cast(float, (unsigned int)(x >> 32)); };
                const float l0 = lo(pq[0]), l1 = lo(pq[1]), l2 = lo(pq[2]), l3 = lo(pq[3]);
                const float h0 = hi(pq[0]), h1 = hi(pq[1]), h2 = hi(pq[2]), h3 = hi(pq[3]);
                const float c0 = f_hi ? h0 : l0, c1 = f_hi ? h1 : l1, c2 = f_hi ? h2 : l2, c3 = f_hi ? h3 : l3;
                const float c4 = c_lo ? l0 : h0, c5 = c_lo ? l1 : h1, c6 = c_lo ? l2 : h2, c7 = c_lo ? l3 : h3;
                // include/interp.h:315-326
                gx = lg_fma(omv, lg_fma(omu, c1 - c0, uu * (c2 - c3)), v * lg_fma(omu, c5 - c4, uu * (c6 - c7)));
                gy = lg_fma(omv, lg_fma(omt, c3 - c0, t * (c2 - c1)), v * lg_fma(omt, c7 - c4, t * (c6 - c5)));
                gz = lg_fma(omu, lg_fma(omt, c4 - c0, t * (c5 - c1)), uu * lg_fma(omt, c7 - c3, t * (c6 - c2)));
            }
            // cuda/interp.cu:230: (Real)((double)diff * dt); for dt = +-1 that is +-diff exactly
            const float diff = UNIT ? (float)dt * gv : (float)((double)gv * dt);
            rux[it] = lg_fma(gx, diff, rux[it]);   // ascending channel order, as the reference's thread-owned sum
            ruy[it] = lg_fma(gy, diff, ruy[it]);
            ruz[it] = lg_fma(gz, diff, ruz[it]);
        }
        // The next channel's grad_out values (requested at the top of this channel) are waited for HERE, in front of the
        // flush: loads and atomics count in one in-order counter, so a wait placed behind the flush -- where the values
        // are consumed -- would also wait for every flush atomic to be acknowledged (~1 us with all CUs flushing)
        // before the next channel's gathers could even be issued.
#pragma unroll
        for (int it = 0; it < VPL; ++it) asm volatile("" : "+v"(pgv[it]));
        __syncthreads();
        // flush touched cells (one wave per window row, lanes along z) and re-zero them for the next channel
        {
            const int lane = threadIdx.x & 63;
            const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
            const uint32_t nrows = (uint32_t)(WX * WY);
            for (uint32_t row = wave; row < nrows; row += NT / 64) {
                const uint32_t lx = sg.d_wy.div(row), ly = row - lx * (uint32_t)WY;
                double *wrow = win + row * (uint32_t)WZ;
                for (int lz = lane; lz < wez; lz += 64) {
                    const double acc = wrow[lz];
                    if (acc != 0.0) {
                        wrow[lz] = 0.0;
                        const int2 o = org[lz >> 4];
                        const uint32_t off = __umul24((uint32_t)clamp1(o.x + (int)lx, nx), gxB) +
                                             __umul24((uint32_t)clamp1(o.y + (int)ly, ny), gyB) + (uint32_t)(wz0 + lz) * 4u;
                        (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32((float)acc, rdI, off, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int it = 0; it < VPL; ++it)
        if (SV[it] != DEAD) {
            st_pol<LAGO_NT_SPLAT_ST>(&dun[SV[it]], rux[it]); st_pol<LAGO_NT_SPLAT_ST>(&dun[SV[it] + nv], ruy[it]); st_pol<LAGO_NT_SPLAT_ST>(&dun[SV[it] + 2 * (size_t)nv], ruz[it]);
        }
}

// TX TY TZ(0 = auto) margins MX MY MZ.  TX is an upper bound: make_shear shrinks it until the float64 window fits
// 80 KB (8 x 6 -> 5 x 6 x 128 at nz = 128, 8 x 6 x 80 at nz = 160: 3840-voxel tiles).  Measured against 4 x 8
// (tools/ab_tiles.py, steady state): 1-3 % faster at 128^3 and 160^3, one and three channels.
static KnobArray<6> g_shear_cfg({8, 6, 0, 1, 1, 4});
// g_shear_mc, several channels with d_u wanted: 0 d_u read-modify-written per channel, 1 d_u in registers over the
// channels (splat_shear_kernel<..., VPL>), 2 (default) the geometry-once kernel (splat_shear_mc_kernel)
static std::atomic<int> g_shear_nt{1024}, g_shear_on{1}, g_shear_mc{2};

// max_vox > 0 (the geometry-once multi-channel kernel): the tile is shrunk further, larger of TX / TY first, until it
// has at most that many voxels.
static bool make_shear(ShearGeom &sg, const Geom &g, int64_t nn, size_t &smem, int max_vox = 0) {
    const std::array<int, 6> cfg = g_shear_cfg.get();
    int TX = cfg[0], TY = cfg[1], TZ = cfg[2];
    const int EX = cfg[3], EY = cfg[4], EZ = cfg[5];
    if (g.nz < 2 || TX < 1 || TY < 1 || EX < 0 || EY < 0 || EZ < 0) return false;
    // strides must fit the 24-bit multiplies
    if ((uint64_t)g.ny * g.nz * 4 >= (1u << 24) || g.nx >= (1 << 23)) return false;
    if (TZ <= 0) {  // auto: whole z rows up to 128 voxels, else even parts of at most 128 (multiples of 16)
        const int parts = (g.nz + 127) / 128;
        TZ = (((g.nz + parts - 1) / parts + 15) / 16) * 16;
    }
    TX = TX < g.nx ? TX : g.nx;
    TY = TY < g.ny ? TY : g.ny;
    TZ = TZ < g.nz ? TZ : g.nz;
    for (;;) {
        sg.WX = TX + 1 + 2 * EX;
        sg.WY = TY + 1 + 2 * EY;
        sg.WZ = TZ >= g.nz ? g.nz : ((TZ + 1 + 2 * EZ + 15 + 15) / 16) * 16;
        if (sg.WZ > g.nz) sg.WZ = g.nz;
        sg.nseg = (sg.WZ + 15) / 16;
        const bool fits = (uint64_t)sg.WX * sg.WY * sg.WZ * 8 + (uint64_t)sg.nseg * 8 <= 80 * 1024;  // two workgroups per CU
        if (fits && (max_vox <= 0 || (int64_t)TX * TY * TZ <= max_vox)) break;
        if (max_vox > 0) {
            if (TX >= TY && TX > 1) --TX;
            else if (TY > 1) --TY;
            else if (TX > 1) --TX;
            else if (TZ > 16) TZ = ((TZ / 2 + 15) / 16) * 16;
            else return false;
        } else if (TX > 2) --TX;
        else if (TY > 2) --TY;
        else if (TZ > 16) TZ = ((TZ / 2 + 15) / 16) * 16;
        else return false;
    }
    if ((uint64_t)sg.WY * sg.WZ * 8 >= (1u << 24) || (int64_t)TX * TY * TZ < 256) return false;
    sg.nx = g.nx; sg.ny = g.ny; sg.nz = g.nz;
    sg.TX = TX; sg.TY = TY; sg.TZ = TZ;
    sg.MX = EX; sg.MY = EY; sg.MZ = EZ;
    sg.win_cells = (uint32_t)sg.WX * sg.WY * sg.WZ;
    smem = (size_t)sg.win_cells * sizeof(double) + (size_t)sg.nseg * 8;
    sg.ntx = (g.nx + TX - 1) / TX;
    sg.nty = (g.ny + TY - 1) / TY;
    sg.ntz = (g.nz + TZ - 1) / TZ;
    sg.tiles_per_item = sg.ntx * sg.nty * sg.ntz;
    const int64_t total = (int64_t)sg.tiles_per_item * nn;
    if (total >= (1ll << 31)) return false;
    sg.total = (uint32_t)total;
    sg.rev = g.rev;
    sg.tile_vox = (uint32_t)TX * TY * TZ;
    sg.d_tiles = FastDiv(sg.tiles_per_item);
    sg.d_tyz = FastDiv(sg.nty * sg.ntz);
    sg.d_tz = FastDiv(sg.ntz);
    sg.d_TyTz = FastDiv((uint32_t)(TY * TZ));
    sg.d_Tz = FastDiv((uint32_t)TZ);
    sg.d_wy = FastDiv((uint32_t)sg.WY);
    return true;
}

template <int NT>
static hipError_t launch_shear(float *d_I, float *d_u, const float *go, const float *I, const float *u, double dt, int nc,
                               const ShearGeom &sg, size_t smem, bool bc, bool need_u, int umode, float addgo, hipStream_t s) {
    const bool unit = unit_dt<float>(dt);
    // several channels with d_u wanted: keep d_u in registers when the workgroup covers the tile in at most 4 passes
    const int passes = (int)((sg.tile_vox + NT - 1) / NT);
    const bool mc = need_u && nc > 1 && passes <= 4 && g_shear_mc;
#define LAGO_SHEAR(NU, UN, B)                                                                                     \
    do {                                                                                                          \
        auto k = !mc ? splat_shear_kernel<NT, NU, UN, B, 0>                                                       \
                     : (passes <= 1 ? splat_shear_kernel<NT, NU, UN, B, (NU ? 1 : 0)>                             \
                        : passes <= 2 ? splat_shear_kernel<NT, NU, UN, B, (NU ? 2 : 0)>                           \
                                      : splat_shear_kernel<NT, NU, UN, B, (NU ? 4 : 0)>);                         \
        if (smem > 64 * 1024) {                                                                                   \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),                                 \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);            \
            if (e != hipSuccess) return e;                                                                        \
        }                                                                                                         \
        hipLaunchKernelGGL(k, dim3(sg.total), dim3(NT), smem, s, d_I, d_u, go, I, u, dt, nc, sg, umode, addgo);   \
    } while (0)
    if (need_u) {
        if (unit) { if (bc) LAGO_SHEAR(true, true, true); else LAGO_SHEAR(true, true, false); }
        else { if (bc) LAGO_SHEAR(true, false, true); else LAGO_SHEAR(true, false, false); }
    } else {
        if (unit) { if (bc) LAGO_SHEAR(false, true, true); else LAGO_SHEAR(false, true, false); }
        else { if (bc) LAGO_SHEAR(false, false, true); else LAGO_SHEAR(false, false, false); }
    }
#undef LAGO_SHEAR
    return hipSuccess;
}

// float32 displacement splat through the sheared-window kernel; returns 1 when the shape is left to the
// general tiled kernel.
static int interp_backward_shear(float *d_I, float *d_u, const float *go, const float *I, const float *u, double dt,
                                 int nc, int64_t nn, const Geom &g, bool bc, bool need_u, int umode, float addgo, hipStream_t s) {
    if (!g_shear_on) return 1;
    ShearGeom sg;
    size_t smem;
    hipError_t e;
    const int shear_nt = g_shear_nt;
    // the geometry-once kernel: in isolation it gains 7-12 % for non-unit steps and measures within +-3 % for dt = +-1
    // (smaller tiles flush more cells); inside lddmm_step -- running d_u / d_I sums, 3-voxel displacements -- it wins
    // for both: 21.63 -> 20.40 ms per step at 8 x 160^3, 10.27 -> 9.85 at 8 x 128^3 (tools/ab_step_mc.py)
    const bool unit_step = unit_dt<float>(dt);
    if (need_u && nc > 1 && g_shear_mc >= 2 && shear_nt >= 1024 && make_shear(sg, g, nn, smem, 2048)) {
        const bool unit = unit_step;
        const bool one = sg.tile_vox <= 1024u;
#define LAGO_SHEAR_MC(UN, B)                                                                                      \
    do {                                                                                                          \
        auto k = one ? splat_shear_mc_kernel<1024, UN, B, 1> : splat_shear_mc_kernel<1024, UN, B, 2>;             \
        if (smem > 64 * 1024) {                                                                                   \
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)smem);                                                                   \
            if (e != hipSuccess) return fail_hip(e, "interp_backward (sheared-window splat)");                    \
        }                                                                                                         \
        hipLaunchKernelGGL(k, dim3(sg.total), dim3(1024), smem, s, d_I, d_u, go, I, u, dt, nc, sg, umode, addgo);  \
    } while (0)
        if (unit) { if (bc) LAGO_SHEAR_MC(true, true); else LAGO_SHEAR_MC(true, false); }
        else { if (bc) LAGO_SHEAR_MC(false, true); else LAGO_SHEAR_MC(false, false); }
#undef LAGO_SHEAR_MC
        note_path(LP_SPLAT_SHEAR_MC);
        return finish_launch(s, "interp_backward (sheared-window splat)");
    }
    if (!make_shear(sg, g, nn, smem)) return 1;
    if (shear_nt >= 1024) e = launch_shear<1024>(d_I, d_u, go, I, u, dt, nc, sg, smem, bc, need_u, umode, addgo, s);
    else if (shear_nt >= 512) e = launch_shear<512>(d_I, d_u, go, I, u, dt, nc, sg, smem, bc, need_u, umode, addgo, s);
    else e = launch_shear<256>(d_I, d_u, go, I, u, dt, nc, sg, smem, bc, need_u, umode, addgo, s);
    if (e != hipSuccess) return fail_hip(e, "interp_backward (sheared-window splat)");
    note_path(LP_SPLAT_SHEAR);
    return finish_launch(s, "interp_backward (sheared-window splat)");
}

// g: target grid; gs: source grid (tiles): the same grid for interp / affine.  sc[d] = how many
// target cells one source step spans along axis d (1 for a displacement field); when given, the
// tile is shrunk until its image fits the LDS window.
static bool make_tiles(TileGeom &tg, const Geom &g, const Geom &gs, int64_t nn, const double *sc, size_t &smem,
                       int &nthreads, const int *cfg_in = nullptr) {
    const std::array<int, 7> cfg_set = g_tile_cfg.get();
    const int *cfg = cfg_in ? cfg_in : cfg_set.data();
    int TX = cfg[0], TY = cfg[1], TZ = cfg[2];
    const int EX = cfg[3], EY = cfg[4], EZ = cfg[5];
    nthreads = cfg[6] >= 1024 ? 1024 : (cfg[6] >= 512 ? 512 : 256);
    if (TY < 1 || EX < 0 || EY < 0 || EZ < 0) return false;
    if (TZ <= 0) {  // auto: whole z rows, split evenly when they are longer than 128 voxels (a longer row
                    // pushes the f64 window of a 4 x 8 tile past 80 KB, i.e. down to one workgroup per CU:
                    // 647 -> 486 us at 8 x 1 x 160^3)
        const int parts = (gs.nz + 127) / 128;
        TZ = (((gs.nz + parts - 1) / parts + 15) / 16) * 16;
    }
    if (TX <= 0) {  // auto: about 4096 voxels per tile (measured optimum at 128^3: 4 x 8 x 128, 512 threads)
        const int per = TY * (TZ < gs.nz ? TZ : gs.nz);
        TX = (4096 + per - 1) / per;
        if (TX < 4) TX = 4;
        // ... but keep the f64 window under 80 KB so that two workgroups fit a CU
        auto wbytes = [&](int tx) {
            const int wz = ((TZ + 1 + 2 * EZ + 15 + 15) / 16) * 16;
            return (size_t)(tx + 1 + 2 * EX) * (TY + 1 + 2 * EY) * (wz < g.nz ? wz : g.nz) * sizeof(double);
        };
        while (TX > 4 && wbytes(TX) > 80 * 1024) --TX;
    }
    TX = TX < gs.nx ? TX : gs.nx;
    TY = TY < gs.ny ? TY : gs.ny;
    TZ = TZ < gs.nz ? TZ : gs.nz;
    tg.nx = g.nx; tg.ny = g.ny; tg.nz = g.nz;
    tg.snx = gs.nx; tg.sny = gs.ny; tg.snz = gs.nz;
    tg.MX = EX; tg.MY = EY; tg.MZ = EZ;
    auto span = [&](int T, int d) {  // target cells covered by T source voxels, + the ceil corner
        if (!sc) return T + 1;
        const double a = sc[d] < 0 ? -sc[d] : sc[d];
        if (!(a < 1e6)) return 1 << 20;
        return (int)((T - 1) * a) + 3;
    };
    for (;;) {
        // window = image of the tile + a margin on both sides of the probed origin
        tg.WX = span(TX, 0) + 2 * EX;
        tg.WY = span(TY, 1) + 2 * EY;
        tg.WZ = ((span(TZ, 2) + 2 * EZ + 15 + 15) / 16) * 16;  // +15: the z origin is aligned down
        if (tg.WX > g.nx) tg.WX = g.nx;
        if (tg.WY > g.ny) tg.WY = g.ny;
        if (tg.WZ > g.nz) tg.WZ = g.nz;
        const uint64_t cells = (uint64_t)tg.WX * tg.WY * tg.WZ;
        if (cells * sizeof(double) <= 160 * 1024) break;
        if (!sc) return false;
        // halve the tile along the axis with the largest window extent
        if (tg.WZ >= tg.WX && tg.WZ >= tg.WY && TZ > 1) TZ = (TZ + 1) / 2;
        else if (tg.WX >= tg.WY && TX > 1) TX = (TX + 1) / 2;
        else if (TY > 1) TY = (TY + 1) / 2;
        else if (TX > 1) TX = (TX + 1) / 2;
        else if (TZ > 1) TZ = (TZ + 1) / 2;
        else return false;
    }
    if ((int64_t)TX * TY * TZ < 256 && (int64_t)gs.nx * gs.ny * gs.nz >= 256) return false;  // not worth a window
    tg.TX = TX; tg.TY = TY; tg.TZ = TZ;
    tg.win_cells = (uint32_t)tg.WX * tg.WY * tg.WZ;
    smem = (size_t)tg.win_cells * sizeof(double);
    tg.ntx = (gs.nx + TX - 1) / TX;
    tg.nty = (gs.ny + TY - 1) / TY;
    tg.ntz = (gs.nz + TZ - 1) / TZ;
    tg.tiles_per_item = tg.ntx * tg.nty * tg.ntz;
    int64_t total = (int64_t)tg.tiles_per_item * nn;
    if (total >= (1ll << 31)) return false;
    tg.total = (uint32_t)total;
    tg.rev = g.rev;
    tg.tile_groups = (uint32_t)TX * TY * TZ;  // voxels per tile
    while (nthreads > 256 && (uint32_t)nthreads > tg.tile_groups) nthreads >>= 1;
    tg.d_tiles = FastDiv(tg.tiles_per_item);
    tg.d_tyz = FastDiv(tg.nty * tg.ntz);
    tg.d_tz = FastDiv(tg.ntz);
    tg.d_TyTzq = FastDiv((uint32_t)(TY * TZ));
    tg.d_Tzq = FastDiv((uint32_t)TZ);
    tg.d_wey = FastDiv((uint32_t)(tg.WY < g.ny ? tg.WY : g.ny));
    return true;
}

std::atomic<int> g_splat_mc{1};  // 1: multi-channel single-pass form of interp_backward where it applies

template <typename R, int MODE, bool BC, bool NEED_U, int NT, int VPL, bool MC = false>
static hipError_t launch_tiled(R *d_I, R *d_u, const R *go, const R *I, const PosArgs &pa, int nc,
                               const TileGeom &tg, size_t smem, hipStream_t s) {
    auto k = splat_tiled_kernel<R, MODE, BC, NEED_U, NT, VPL, MC>;
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3(tg.total), dim3(NT), smem, s, d_I, d_u, go, I, pa, nc, tg);
    return hipSuccess;
}

template <typename R, int MODE, bool BC, bool NEED_U, int VPL>
static hipError_t by_threads(R *d_I, R *d_u, const R *go, const R *I, const PosArgs &pa, int nc, const TileGeom &tg,
                             size_t smem, int nt, hipStream_t s) {
    if (nt >= 1024) return launch_tiled<R, MODE, BC, NEED_U, 1024, VPL>(d_I, d_u, go, I, pa, nc, tg, smem, s);
    if (nt >= 512) return launch_tiled<R, MODE, BC, NEED_U, 512, VPL>(d_I, d_u, go, I, pa, nc, tg, smem, s);
    return launch_tiled<R, MODE, BC, NEED_U, 256, VPL>(d_I, d_u, go, I, pa, nc, tg, smem, s);
}

// Returns LAGO_OK / error, or 1 when this shape is left to the plain kernel.
// The caller has already zeroed d_I (and d_u when it is not needed).
template <typename R>
int interp_backward_lds(R *d_I, R *d_u, const R *go, const R *I, const R *u, double dt, int nc, int64_t nn,
                        const Geom &g, bool bc, bool need_u, int umode, double addgo, hipStream_t s) {
    constexpr int V = 4;
    const bool vec = g_interp_vec != 0;
    TileGeom tg;
    size_t smem;
    int nt;
    if (g.nz < 2) return 1;  // thin volumes take the plain kernel
    if constexpr (sizeof(R) == 4) {
        if (vec) {
            const int rc = interp_backward_shear((float *)d_I, (float *)d_u, (const float *)go, (const float *)I,
                                                 (const float *)u, dt, nc, nn, g, bc, need_u, umode, (float)addgo, s);
            if (rc != 1) return rc;
        }
    }
    if (!make_tiles(tg, g, g, nn, nullptr, smem, nt)) return 1;
    PosArgs pa{};
    pa.u = u;
    pa.dt = dt;
    pa.umode = umode;
    pa.addgo = addgo;
    hipError_t e;
    const bool unit = unit_dt<R>(dt);
    // several channels with d_u wanted: the single-pass multi-channel form when one workgroup pass
    // covers the tile (1024 threads x 4 voxels for the default 4096-voxel tile)
    // (float32 only: the float64 instantiations of this form need 125 spilled registers at 128 VGPRs; float64 takes the
    // per-channel passes below)
    if constexpr (sizeof(R) == 4)
    if (g_splat_mc && vec && need_u && nc > 1 && tg.tile_groups <= 1024u * V) {
        const int ntm = tg.tile_groups <= 256u * V ? 256 : (tg.tile_groups <= 512u * V ? 512 : 1024);
#define GOMC(M, B)                                                                                                 \
    e = ntm == 1024  ? launch_tiled<R, M, B, true, 1024, V, true>(d_I, d_u, go, I, pa, nc, tg, smem, s)           \
        : ntm == 512 ? launch_tiled<R, M, B, true, 512, V, true>(d_I, d_u, go, I, pa, nc, tg, smem, s)            \
                     : launch_tiled<R, M, B, true, 256, V, true>(d_I, d_u, go, I, pa, nc, tg, smem, s)
        if (unit) {
            if (bc) GOMC(POS_DISP_UNIT, true); else GOMC(POS_DISP_UNIT, false);
        } else {
            if (bc) GOMC(POS_DISP, true); else GOMC(POS_DISP, false);
        }
#undef GOMC
        if (e != hipSuccess) return fail_hip(e, "interp_backward (tiled splat)");
        note_path(LP_SPLAT_TILED);
        return finish_launch(s, "interp_backward (tiled splat)");
    }
#define GO(B, U) \
    e = !vec   ? by_threads<R, POS_DISP, B, U, 1>(d_I, d_u, go, I, pa, nc, tg, smem, nt, s)          \
        : unit ? by_threads<R, POS_DISP_UNIT, B, U, V>(d_I, d_u, go, I, pa, nc, tg, smem, nt, s)     \
               : by_threads<R, POS_DISP, B, U, V>(d_I, d_u, go, I, pa, nc, tg, smem, nt, s)
    if (bc) {
        if (need_u) GO(true, true); else GO(true, false);
    } else {
        if (need_u) GO(false, true); else GO(false, false);
    }
#undef GO
    if (e != hipSuccess) return fail_hip(e, "interp_backward (tiled splat)");
    note_path(LP_SPLAT_TILED);
    return finish_launch(s, "interp_backward (tiled splat)");
}

// Image splat of affine_interp_backward (cuda/affine.cu:330-536, the d_I part), 3D.
// d_I already zeroed.  Same return convention as interp_backward_lds.
template <typename R>
int affine_splat_lds(R *d_I, const R *go, const R *A, const R *T, int nc, int64_t nn, const Geom &g, bool bc,
                     hipStream_t s, int gate) {
    TileGeom tg;
    size_t smem;
    int nt;
    if (g.nz < 2) return 1;
    // a sheared / rotated tile spreads along every axis: compact tiles keep its image inside the
    // window (the thin 4 x 8 x nz default of the displacement mode does not)
    static const int affine_cfg[7] = {16, 8, 64, 1, 1, 4, 1024};
    if (!make_tiles(tg, g, g, nn, nullptr, smem, nt, affine_cfg)) return 1;
    PosArgs pa{};
    pa.A = A;
    pa.T = T;
    pa.gate = gate;
    hipError_t e = bc ? by_threads<R, POS_AFFINE, true, false, 4>(d_I, nullptr, go, nullptr, pa, nc, tg, smem, nt, s)
                      : by_threads<R, POS_AFFINE, false, false, 4>(d_I, nullptr, go, nullptr, pa, nc, tg, smem, nt, s);
    if (e != hipSuccess) return fail_hip(e, "affine_interp_backward (tiled splat)");
    return finish_launch(s, "affine_interp_backward (tiled splat)");
}

// regrid_backward (cuda/affine.cu:767-855), 3D: every (n, c) plane of grad_out (source grid gs)
// is one batch item with a single channel.  d_I (target grid g) already zeroed.
template <typename R>
int regrid_splat_lds(R *d_I, const R *go, int64_t nplanes, const Geom &g, const Geom &gs, const double *O,
                     const double *S, hipStream_t s) {
    TileGeom tg;
    size_t smem;
    int nt;
    if (g.nz < 2 || gs.nz < 2) return 1;
    if (!make_tiles(tg, g, gs, nplanes, S, smem, nt)) return 1;
    note_path(LP_SPLAT_TILED);
    PosArgs pa{};
    for (int d = 0; d < 3; ++d) {
        pa.O[d] = O[d];
        pa.S[d] = S[d];
    }
    hipError_t e = by_threads<R, POS_REGRID, false, false, 4>(d_I, nullptr, go, nullptr, pa, 1, tg, smem, nt, s);
    if (e != hipSuccess) return fail_hip(e, "regrid_backward (tiled splat)");
    return finish_launch(s, "regrid_backward (tiled splat)");
}

template int interp_backward_lds<float>(float *, float *, const float *, const float *, const float *, double, int,
                                        int64_t, const Geom &, bool, bool, int, double, hipStream_t);
template int interp_backward_lds<double>(double *, double *, const double *, const double *, const double *, double,
                                         int, int64_t, const Geom &, bool, bool, int, double, hipStream_t);
template int affine_splat_lds<float>(float *, const float *, const float *, const float *, int, int64_t, const Geom &,
                                     bool, hipStream_t, int);
template int affine_splat_lds<double>(double *, const double *, const double *, const double *, int, int64_t,
                                      const Geom &, bool, hipStream_t, int);
template int regrid_splat_lds<float>(float *, const float *, int64_t, const Geom &, const Geom &, const double *,
                                     const double *, hipStream_t);
template int regrid_splat_lds<double>(double *, const double *, int64_t, const Geom &, const Geom &, const double *,
                                      const double *, hipStream_t);

}  // namespace lago

namespace lago {
void tune_splat(const int32_t *tile7, const int32_t *shear8, int shear_mc, int mc) {
    g_tile_cfg.set({tile7[0], tile7[1], tile7[2], tile7[3], tile7[4], tile7[5], tile7[6]});
    g_shear_on = shear8[0];
    g_shear_cfg.set({shear8[1], shear8[2], shear8[3], shear8[4], shear8[5], shear8[6]});
    g_shear_nt = shear8[7];
    g_shear_mc = shear_mc;
    g_splat_mc = mc;
}
}  // namespace lago
