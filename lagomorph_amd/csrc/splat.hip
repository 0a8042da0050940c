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

namespace lago {

#ifdef LAGO_PROFILING   // profiling builds only: per-wave shader-clock stamps at phase boundaries (tools/stamp_splat.py)
__device__ unsigned long long *g_dev_stamps = nullptr;   // [workgroup][wave 0..15][8]
#define LAGO_STAMP(i)                                                                                             \
    do {                                                                                                          \
        if (g_dev_stamps && (threadIdx.x & 63u) == 0)                                                             \
            g_dev_stamps[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define LAGO_STAMP(i) \
    do {              \
    } while (0)
#endif

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
    // image window of splat_shear_iw_kernel: the same cells as float32, in 16-byte chunks of four z cells
    uint32_t iw_chunks, iw_bytes;   // chunks; bytes reserved (whole 1 KB wave-instructions)
    FastDiv d_wzc;                  // chunks per window row (WZ / 4)
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

    LAGO_STAMP(0);
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
        LAGO_STAMP(1);
        for (uint32_t f = threadIdx.x; f < sg.win_cells; f += NT) win[f] = 0.0;
        LAGO_STAMP(2);
        __syncthreads();
        LAGO_STAMP(3);
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
            const float gv = gc[sv];
            const float hx = shear_pos<UNIT>(vi, dt, un[sv]);
            const float hy = shear_pos<UNIT>(vj, dt, un[sv + nv]);
            const float hz = shear_pos<UNIT>(vk, dt, un[sv + 2 * (size_t)nv]);
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
                if ((uint32_t)fx < (uint32_t)(nx - 1) && (uint32_t)fy < (uint32_t)(ny - 1)) {
                    // rows unclamped: (fx,fy) (fx+1,fy) (fx+1,fy+1) (fx,fy+1); the z pair is fetched at
                    // zb = clamp(fz, 0, nz-2) and picked as Lerp3 does (common.hpp): beyond the upper border both
                    // corners are the pair's high half, below the lower border both are its low half
                    const int zb = lg_med3(fz, 0, nz - 2);
                    const bool f_hi = fz > nz - 2, c_lo = fz < 0;
                    const uint32_t o = __umul24((uint32_t)fx, gxB) + __umul24((uint32_t)fy, gyB) + (uint32_t)zb * 4u;
                    float l0, l1, l2, l3, h0, h1, h2, h3;
                    buf_load2s(rI, o, 0u, l0, h0);          // (the full-width-integer form: see common.hpp on the
                    buf_load2s(rI, o, gxB, l1, h1);         //  hipcc narrowing of b64 buffer loads)
                    buf_load2s(rI, o, gxB + gyB, l2, h2);
                    buf_load2s(rI, o, gyB, l3, h3);
                    const float c0 = f_hi ? h0 : l0, c1 = f_hi ? h1 : l1, c2 = f_hi ? h2 : l2, c3 = f_hi ? h3 : l3;
                    const float c4 = c_lo ? l0 : h0, c5 = c_lo ? l1 : h1, c6 = c_lo ? l2 : h2, c7 = c_lo ? l3 : h3;
                    // include/interp.h:315-326
                    gx = lg_fma(omv, lg_fma(omu, c1 - c0, uu * (c2 - c3)), v * lg_fma(omu, c5 - c4, uu * (c6 - c7)));
                    gy = lg_fma(omv, lg_fma(omt, c3 - c0, t * (c2 - c1)), v * lg_fma(omt, c7 - c4, t * (c6 - c5)));
                    gz = lg_fma(omu, lg_fma(omt, c4 - c0, t * (c5 - c1)), uu * lg_fma(omt, c7 - c3, t * (c6 - c2)));
                } else {
                    Lerp3<float, false> Lq;
                    Lq.setup(hx, hy, hz, nx, ny, nz);
                    Lq.grad(Ic, gx, gy, gz);
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
                    dun[sv] = ix; dun[sv + nv] = iy; dun[sv + 2 * (size_t)nv] = iz;
                }
            }
        }
        LAGO_STAMP(4);
        __syncthreads();
        LAGO_STAMP(5);
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
        LAGO_STAMP(6);
        __syncthreads();
        LAGO_STAMP(7);
    }
}

// ---------------------------------------------------------------------------------------------
// Row-mapped form of the sheared-window splat (round 4): the SAME algorithm as splat_shear_kernel with the index
// arithmetic taken out of the voxel loop.  The ablation of the image-window kernels (profiles/r04_splat_pipeline.md)
// showed these kernels to be bound by instruction issue, not by memory: with every load, store, atomic and LDS
// operation switched off the skeleton still took 146 of 205 us.  splat_shear_kernel decodes every voxel from a running
// tile index (two multiply-high divisions, 64-bit offsets) and rebuilds its plane offset from (i, j, k).  Here a
// workgroup is TY x TZ threads -- one per (y, z) of the tile -- and the passes of its loop are the tile's x slabs: a
// voxel's offset in a plane is a per-thread constant plus a SCALAR slab offset that rides in the soffset field of the
// buffer instructions, (y, z) and their float conversions are per-thread constants, x is wave-uniform.  Everything
// else -- positions, sequentially flipped weights, window addressing, the clamped fall-backs, the gradient
// expressions, the flush -- is splat_shear_kernel's, expression by expression: d_u bit-identical.
template <bool NEED_U, bool UNIT, bool BC>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8, 8))) void splat_row_kernel(float *__restrict__ d_I, float *__restrict__ d_u,
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
    // window z range: 16-aligned start (flush rows start on 64-byte boundaries), inside the grid; a window as long as the
    // z rows starts at 0 whatever the displacement (no probe: one dependent round trip less)
    const int cxs = x0 + ex / 2, cys = y0 + ey / 2;
    int wz0 = 0;
    if (wez < nz) {
        const float fdt = (float)dt;
        const size_t sc = ((size_t)cxs * ny + cys) * nz + (z0 + ez / 2);
        const int bzo = z0 + (int)floorf(fdt * un[sc + 2 * (size_t)nv]);
        wz0 = max(0, min((bzo - sg.MZ) & ~15, nz - wez));
    }
    // origin of every z segment: the displacement of the tile's centre column at the segment's height -- requested
    // first, consumed behind the zeroing of the window (their round trip hides under it)
    float pox = 0.f, poy = 0.f;
    if ((int)threadIdx.x < sg.nseg) {
        const int zc = min(wz0 + (int)threadIdx.x * 16 + 8, nz - 1);
        const size_t sc = ((size_t)cxs * ny + cys) * nz + zc;
        pox = un[sc];
        poy = un[sc + nv];
    }
    const uint32_t sxB = (uint32_t)(WY * WZ) * 8u, syB = (uint32_t)WZ * 8u;            // window strides in bytes
    const uint32_t gxB = (uint32_t)ny * nz * 4u, gyB = (uint32_t)nz * 4u;              // grid strides in bytes
    const uint32_t wxu1 = (uint32_t)(WX - 1), wyu1 = (uint32_t)(WY - 1), wezu = (uint32_t)wez;
    const uint32_t NT = blockDim.x;   // = TY * TZ
    // ---- this thread's (y, z) of the tile, once: the passes below walk the tile's x slabs, so a voxel's plane offset is
    // a per-thread constant plus a SCALAR slab offset (the instruction's soffset field): no per-voxel index arithmetic
    const uint32_t tb = sg.d_Tz.div(threadIdx.x), tk = threadIdx.x - tb * (uint32_t)sg.TZ;
    const bool live = (int)tb < ey && (int)tk < ez;
    const int vj = y0 + (int)tb, vk = z0 + (int)tk;
    const uint32_t toff = live ? ((uint32_t)vj * (uint32_t)nz + (uint32_t)vk) * 4u : 0x80000000u;   // (beyond any plane: reads 0, stores dropped)
    const BufRsrc ru0 = make_rsrc(un, planeB), ru1 = make_rsrc(un + nv, planeB), ru2 = make_rsrc(un + 2 * (size_t)nv, planeB);
    for (int c = 0; c < nc; ++c) {
        {   // (16-byte stores; win_cells is even: WZ is)
            double2 *w2 = reinterpret_cast<double2 *>(win);
            for (uint32_t f = threadIdx.x; f < sg.win_cells / 2; f += NT) w2[f] = make_double2(0.0, 0.0);
            if (threadIdx.x == 0 && (sg.win_cells & 1u)) win[sg.win_cells - 1] = 0.0;
        }
        if (c == 0 && (int)threadIdx.x < sg.nseg) {
            const float fdt = (float)dt;
            int2 o;
            o.x = max(-1, min(x0 + (int)floorf(fdt * pox) - sg.MX, nx + 1 - WX));
            o.y = max(-1, min(y0 + (int)floorf(fdt * poy) - sg.MY, ny + 1 - WY));
            org[threadIdx.x] = o;
        }
        __syncthreads();
        const float *Ic = In + (size_t)c * nv;
        const float *gc = gon + (size_t)c * nv;
        const BufRsrc rdI = make_rsrc(dIn + (size_t)c * nv, planeB);
        const BufRsrc rI = make_rsrc(Ic, planeB);
        const BufRsrc rd0 = make_rsrc(dun, planeB), rd1 = make_rsrc(NEED_U ? dun + nv : dun, planeB), rd2 = make_rsrc(NEED_U ? dun + 2 * (size_t)nv : dun, planeB);
        const BufRsrc rg0 = make_rsrc(gon, planeB), rg1 = make_rsrc(gon + (nc == 3 ? nv : 0u), planeB), rg2 = make_rsrc(gon + (nc == 3 ? 2 * (size_t)nv : 0u), planeB);
        const BufRsrc rgo = make_rsrc(gc, planeB);
        for (int it = 0; it < ex; ++it) {
            const int vi = x0 + it;
            const uint32_t soff = (uint32_t)vi * gxB;   // scalar
            const float gv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rgo, toff, soff, 0));
            const float ux = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ru0, toff, soff, 0));
            const float uy = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ru1, toff, soff, 0));
            const float uz = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ru2, toff, soff, 0));
            if (!live) continue;
            const float hx = shear_pos<UNIT>(vi, dt, ux);
            const float hy = shear_pos<UNIT>(vj, dt, uy);
            const float hz = shear_pos<UNIT>(vk, dt, uz);
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
                if ((uint32_t)fx < (uint32_t)(nx - 1) && (uint32_t)fy < (uint32_t)(ny - 1)) {
                    // rows unclamped: (fx,fy) (fx+1,fy) (fx+1,fy+1) (fx,fy+1); the z pair is fetched at
                    // zb = clamp(fz, 0, nz-2) and picked as Lerp3 does (common.hpp): beyond the upper border both
                    // corners are the pair's high half, below the lower border both are its low half
                    const int zb = lg_med3(fz, 0, nz - 2);
                    const bool f_hi = fz > nz - 2, c_lo = fz < 0;
                    const uint32_t o = __umul24((uint32_t)fx, gxB) + __umul24((uint32_t)fy, gyB) + (uint32_t)zb * 4u;
                    float l0, l1, l2, l3, h0, h1, h2, h3;
                    buf_load2s(rI, o, 0u, l0, h0);          // (the full-width-integer form: see common.hpp on the
                    buf_load2s(rI, o, gxB, l1, h1);         //  hipcc narrowing of b64 buffer loads)
                    buf_load2s(rI, o, gxB + gyB, l2, h2);
                    buf_load2s(rI, o, gyB, l3, h3);
                    const float c0 = f_hi ? h0 : l0, c1 = f_hi ? h1 : l1, c2 = f_hi ? h2 : l2, c3 = f_hi ? h3 : l3;
                    const float c4 = c_lo ? l0 : h0, c5 = c_lo ? l1 : h1, c6 = c_lo ? l2 : h2, c7 = c_lo ? l3 : h3;
                    // include/interp.h:315-326
                    gx = lg_fma(omv, lg_fma(omu, c1 - c0, uu * (c2 - c3)), v * lg_fma(omu, c5 - c4, uu * (c6 - c7)));
                    gy = lg_fma(omv, lg_fma(omt, c3 - c0, t * (c2 - c1)), v * lg_fma(omt, c7 - c4, t * (c6 - c5)));
                    gz = lg_fma(omu, lg_fma(omt, c4 - c0, t * (c5 - c1)), uu * lg_fma(omt, c7 - c3, t * (c6 - c2)));
                } else {
                    Lerp3<float, false> Lq;
                    Lq.setup(hx, hy, hz, nx, ny, nz);
                    Lq.grad(Ic, gx, gy, gz);
                }
                // cuda/interp.cu:230: (Real)((double)diff * dt); for dt = +-1 that is +-diff exactly
                const float diff = UNIT ? (float)dt * gv : (float)((double)gv * dt);
                // ascending channel order, as the reference's thread-owned accumulation; the start value is zero
                // for the reference operator, the caller's d_u or addgo * grad_out for the fused backward forms
                float ix = 0.f, iy = 0.f, iz = 0.f;
                if (c > 0 || umode == 1) {
                    ix = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd0, toff, soff, 0));
                    iy = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd1, toff, soff, 0));
                    iz = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd2, toff, soff, 0));
                } else if (umode == 2) {
                    ix = addgo * __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg0, toff, soff, 0));
                    iy = addgo * __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg1, toff, soff, 0));
                    iz = addgo * __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg2, toff, soff, 0));
                }
                ix = lg_fma(gx, diff, ix);
                iy = lg_fma(gy, diff, iy);
                iz = lg_fma(gz, diff, iz);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, ix), rd0, toff, soff, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, iy), rd1, toff, soff, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, iz), rd2, toff, soff, 0);
            }
        }
        __syncthreads();
        // flush touched cells: one wave per window row (lx, ly), lanes along z; the (x, y) a cell belongs to is
        // its segment's origin + (lx, ly), folded onto the grid (the clamp of the reference).  A row's cells are read
        // first (independent LDS reads), then added.
        {
            const int lane = threadIdx.x & 63;
            const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
            const uint32_t nrows = (uint32_t)(WX * WY);
            const uint32_t nwaves = NT >> 6;
            constexpr int FP = 3;   // 64-cell parts of a window row (WZ <= 192; longer rows take the loop below)
            int2 so[FP];
#pragma unroll
            for (int j = 0; j < FP; ++j) so[j] = org[min((uint32_t)(lane + 64 * j) >> 4, (uint32_t)sg.nseg - 1u)];
            if (wez <= 64 * FP) {
                for (uint32_t row = wave; row < nrows; row += 2 * nwaves) {
                    const uint32_t row2 = row + nwaves;
                    double acc[2][FP];
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int j = 0; j < FP; ++j) {
                            const int lz = lane + 64 * j;
                            const uint32_t rw = r ? row2 : row;
                            acc[r][j] = (rw < nrows && lz < wez) ? win[rw * (uint32_t)WZ + lz] : 0.0;
                        }
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const uint32_t rw = r ? row2 : row;
                        const uint32_t lx = sg.d_wy.div(rw), ly = rw - lx * (uint32_t)WY;
#pragma unroll
                        for (int j = 0; j < FP; ++j) {
                            if (acc[r][j] != 0.0) {
                                const uint32_t off = __umul24((uint32_t)clamp1(so[j].x + (int)lx, nx), gxB) +
                                                     __umul24((uint32_t)clamp1(so[j].y + (int)ly, ny), gyB) +
                                                     (uint32_t)(wz0 + lane + 64 * j) * 4u;
                                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32((float)acc[r][j], rdI, off, 0, 0);
                            }
                        }
                    }
                }
            } else {
                for (uint32_t row = wave; row < nrows; row += nwaves) {
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
        }
        if (c + 1 < nc) __syncthreads();   // (the next channel zeroes the window; nothing follows the last flush)
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
//   flags in its low bits (or NOROW), the three fractions, the voxel index and the three d_u sums
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
        const float hx = shear_pos<UNIT>(vi, dt, un[sv]);
        const float hy = shear_pos<UNIT>(vj, dt, un[sv + nv]);
        const float hz = shear_pos<UNIT>(vk, dt, un[sv + 2 * (size_t)nv]);
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
        if ((uint32_t)fx < (uint32_t)(nx - 1) && (uint32_t)fy < (uint32_t)(ny - 1)) {
            // rows unclamped; the z pair is fetched at zb = clamp(fz, 0, nz-2) and picked as Lerp3 does (common.hpp)
            const int zb = lg_med3(fz, 0, nz - 2);
            const uint32_t f_hi = fz > nz - 2 ? 1u : 0u, c_lo = fz < 0 ? 2u : 0u;
            OG[it] = (__umul24((uint32_t)fx, gxB) + __umul24((uint32_t)fy, gyB) + (uint32_t)zb * 4u) | f_hi | c_lo;
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
        for (int it = 0; it < VPL; ++it) pgv[it] = buf_load1<float>(rg, SV[it] == DEAD ? DEAD : SV[it] * 4u);
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
            {
                const uint32_t o = OG[it] & ~3u;
                pq[0] = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(rI, o, 0u, 0));
                pq[1] = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(rI, o, gxB, 0));
                pq[2] = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(rI, o, gxB + gyB, 0));
                pq[3] = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(rI, o, gyB, 0));
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
            const uint32_t a0 = A0[it], a1 = A1[it];
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
                if (OG[it] != NOROW) {
                    // rows unclamped: the cells follow from the gather offset (fx, fy, zb) and its z-border flags
                    const uint32_t o = OG[it] & ~3u;
                    X0 = o; X1 = o + gxB; Y0 = 0u; Y1 = gyB;
                    Z0 = (OG[it] & 1u) ? 4u : 0u;              // floor beyond the upper face: both cells are nz - 1 = zb + 1
                    Z1 = (OG[it] & 2u) ? 0u : 4u;              // floor below the lower face: both cells are 0 = zb
                } else {
                    uint32_t tt = threadIdx.x + (uint32_t)it * NT;
                    // opaque: otherwise the coordinates of this rare path, converted to double, are hoisted out of the
                    // channel loop and SPILLED by every lane (ten dwords: 335 MB of scratch writes per launch at 8 x 3 x 128^3)
                    asm volatile("" : "+v"(tt));
                    const uint32_t a = sg.d_TyTz.div(tt);
                    const uint32_t rr = tt - a * (uint32_t)(sg.TY * sg.TZ);
                    const uint32_t b = sg.d_Tz.div(rr);
                    const uint32_t kk = rr - b * (uint32_t)sg.TZ;
                    const int fx = lg_floor(shear_pos<UNIT>(x0 + (int)a, dt, un[sv]));
                    const int fy = lg_floor(shear_pos<UNIT>(y0 + (int)b, dt, un[sv + nv]));
                    const int fz = lg_floor(shear_pos<UNIT>(z0 + (int)kk, dt, un[sv + 2 * (size_t)nv]));
                    X0 = __umul24((uint32_t)clamp1(fx, nx), gxB); X1 = __umul24((uint32_t)clamp1(fx + 1, nx), gxB);
                    Y0 = __umul24((uint32_t)clamp1(fy, ny), gyB); Y1 = __umul24((uint32_t)clamp1(fy + 1, ny), gyB);
                    Z0 = (uint32_t)clamp1(fz, nz) * 4u; Z1 = (uint32_t)clamp1(fz + 1, nz) * 4u;
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
            if (OG[it] != NOROW) {
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
            } else {  // rows clamped (a sample outside the grid): position again, same expressions, same bits
                uint32_t tt = threadIdx.x + (uint32_t)it * NT;
                asm volatile("" : "+v"(tt));  // as above: nothing of this path may be hoisted out of the channel loop
                const uint32_t a = sg.d_TyTz.div(tt);
                const uint32_t rr = tt - a * (uint32_t)(sg.TY * sg.TZ);
                const uint32_t b = sg.d_Tz.div(rr);
                const uint32_t kk = rr - b * (uint32_t)sg.TZ;
                Lerp3<float, false> Lq;
                Lq.setup(shear_pos<UNIT>(x0 + (int)a, dt, un[sv]), shear_pos<UNIT>(y0 + (int)b, dt, un[sv + nv]),
                         shear_pos<UNIT>(z0 + (int)kk, dt, un[sv + 2 * (size_t)nv]), nx, ny, nz);
                Lq.grad(Ic, gx, gy, gz);
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
            dun[SV[it]] = rux[it]; dun[SV[it] + nv] = ruy[it]; dun[SV[it] + 2 * (size_t)nv] = ruz[it];
        }
}

// ---------------------------------------------------------------------------------------------
// Sheared-window splat whose d_u gathers go through an LDS window of I (round 4).
//
// The kernels above take the eight corners of the d_u term (include/interp.h:207-327) with four pair gathers per
// voxel-channel through the vector L1 -- 70 of the 190 us of the C = 1 kernel at 8 x 128^3, and DEPENDENT round trips
// inside a barrier-phased workgroup: u, grad_out -> position -> gathers -> d_u, pass after pass.  Here the footprint
// cells of the tile exist twice in LDS: the float64 accumulation window (as above) and, behind it, the SAME cells of
// I as float32 (12 instead of 8 bytes per cell), filled with LDS-direct `buffer_load_dwordx4 ... lds` in coalesced
// 16-byte chunks -- each 16-cell z segment from its own sheared (x, y) origin, rows clamped to the grid exactly as
// the reference clamps its corner indices.  One LDS address pair per voxel serves both windows: the corners for the
// gradient are eight ds_read_b32 at (A0, A1)/2 + the row strides, the contributions eight ds_add_f64 at A0, A1 + ....
// Everything a tile needs from global memory is requested up front -- u, grad_out and the d_u start values of all its
// voxels (registers), then the image window -- so a tile costs ONE exposed memory round trip instead of two per pass;
// the accumulation window is zeroed underneath it.  Per channel: [wait, barrier] corners + adds [barrier] next
// channel's image window requested, flush.  Tiles are at most VPL x 1024 voxels (geometry, fractions and d_u sums in
// registers over the channel loop, as splat_shear_mc_kernel).  Samples whose footprint leaves the window (well
// under 1 % of a smooth field) recompute their position and take the reference's clamped global atomics and
// Lerp3's pair gathers.  Arithmetic per voxel and channel is that of splat_shear_kernel: d_u bit-identical.
template <int NT, bool UNIT, bool BC, int VPL, int WPE>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(WPE, WPE))) void splat_shear_iw_kernel(
    float *__restrict__ d_I, float *__restrict__ d_u, const float *__restrict__ go, const float *__restrict__ I,
    const float *__restrict__ u, double dt, int nc, ShearGeom sg, int umode, float addgo) {
    extern __shared__ __align__(16) unsigned char lago_smem[];
    double *win = reinterpret_cast<double *>(lago_smem);
    const uint32_t iwB = sg.win_cells * 8u;                                   // byte offset of the image window
    const int nx = sg.nx, ny = sg.ny, nz = sg.nz;
    const uint32_t nv = (uint32_t)nx * ny * nz;
    const uint32_t planeB = nv * 4u;
    constexpr uint32_t NOWIN = 0xffffffffu, DEAD = 0xffffffffu;
    constexpr int RCH = 2;   // image-window chunks per thread (host: iw_chunks <= RCH * NT)

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

    // ---- window placement first: the two probe loads of a z segment's origin are the OLDEST loads of the wave, so the
    // wait in front of barrier 1 does not cover the voxel operands requested next (loads return in order)
    const int WX = sg.WX, WY = sg.WY, WZ = sg.WZ;
    const int wez = min(WZ, nz);
    const int cxs = x0 + ex / 2, cys = y0 + ey / 2;
    int wz0 = 0;
    if (wez < nz) {   // (a window as long as the z rows starts at 0 whatever the displacement)
        const float fdt = (float)dt;
        const size_t sc = ((size_t)cxs * ny + cys) * nz + (z0 + ez / 2);
        const int bzo = z0 + (int)floorf(fdt * un[sc + 2 * (size_t)nv]);   // (wave-uniform address: a scalar load)
        wz0 = max(0, min((bzo - sg.MZ) & ~15, nz - wez));
    }
    // (x, y) origin per 16-cell z segment: lane s < nseg of EVERY wave probes the displacement of the tile's centre column
    // at segment s and keeps the origin in a register; a look-up is one ds_bpermute (the LDS crossbar, not LDS memory:
    // a table in LDS would be read behind the LDS-direct loads below, and hipcc makes any LDS read wait for them)
    float pox, poy;
    {
        const int lane = (int)(threadIdx.x & 63u);
        const int zc = min(wz0 + lane * 16 + 8, nz - 1);
        const uint32_t off = lane < sg.nseg ? (((uint32_t)cxs * ny + cys) * nz + zc) * 4u : DEAD;
        pox = buf_load1<float>(make_rsrc(un, planeB), off);
        poy = buf_load1<float>(make_rsrc(un + nv, planeB), off);
    }
    // ---- every voxel operand of the tile, requested at once (out-of-range offsets -- DEAD -- read 0)
    uint32_t SV[VPL];
    float ux[VPL], uy[VPL], uz[VPL], pgv[VPL], rux[VPL], ruy[VPL], ruz[VPL];
    {
        const BufRsrc rux_ = make_rsrc(un, planeB), ruy_ = make_rsrc(un + nv, planeB), ruz_ = make_rsrc(un + 2 * (size_t)nv, planeB);
        const BufRsrc rg = make_rsrc(gon, planeB);
#pragma unroll
        for (int it = 0; it < VPL; ++it) {
            const uint32_t tt = threadIdx.x + (uint32_t)it * NT;
            const uint32_t a = sg.d_TyTz.div(tt);
            const uint32_t rr = tt - a * (uint32_t)(sg.TY * sg.TZ);
            const uint32_t b = sg.d_Tz.div(rr);
            const uint32_t kk = rr - b * (uint32_t)sg.TZ;
            const bool live = tt < sg.tile_vox && (int)a < ex && (int)b < ey && (int)kk < ez;
            SV[it] = live ? (((uint32_t)(x0 + a) * ny + (y0 + b)) * nz + (z0 + kk)) * 4u : DEAD;   // byte offset in a plane
            ux[it] = buf_load1<float>(rux_, SV[it]);
            uy[it] = buf_load1<float>(ruy_, SV[it]);
            uz[it] = buf_load1<float>(ruz_, SV[it]);
            pgv[it] = buf_load1<float>(rg, SV[it]);
        }
    }
    {   // zero the accumulation window under the latency of the loads above (16-byte stores; win_cells is even)
        double2 *w2 = reinterpret_cast<double2 *>(win);
        for (uint32_t f = threadIdx.x; f < sg.win_cells / 2; f += NT) w2[f] = make_double2(0.0, 0.0);
    }
    int orgp;   // this lane's segment origin, packed (x + 1) << 16 | (y + 1) (host: nx, ny < 32768)
    {
        const float fdt = (float)dt;
        const int ox = max(-1, min(x0 + (int)floorf(fdt * pox) - sg.MX, nx + 1 - WX));
        const int oy = max(-1, min(y0 + (int)floorf(fdt * poy) - sg.MY, ny + 1 - WY));
        orgp = ((ox + 1) << 16) | (oy + 1);
    }
    auto org_of = [&](uint32_t seg) {   // seg < nseg <= 64
        const int p = __builtin_amdgcn_ds_bpermute((int)(seg << 2), orgp);
        int2 o;
        o.x = (p >> 16) - 1;
        o.y = (p & 0xffff) - 1;
        return o;
    };
    const uint32_t sxB = (uint32_t)(WY * WZ) * 8u, syB = (uint32_t)WZ * 8u;            // window strides in bytes
    const uint32_t gxB = (uint32_t)ny * nz * 4u, gyB = (uint32_t)nz * 4u;              // grid strides in bytes
    const uint32_t wxu1 = (uint32_t)(WX - 1), wyu1 = (uint32_t)(WY - 1), wezu = (uint32_t)wez;

    // ---- image window: chunk c of the window (four z cells; row-major [lx][ly][z/4]) comes from the clamped grid row of
    // its segment's origin; recomputed per channel (two registers kept over the channel loop cost a spill)
    auto request_image = [&](int c) {
        const BufRsrc rI = make_rsrc(In + (size_t)c * nv, planeB);
#pragma unroll
        for (int q = 0; q < RCH; ++q) {
            if ((uint32_t)(q * NT) + (threadIdx.x & ~63u) >= sg.iw_chunks) break;   // wave-uniform
            uint32_t ch = (uint32_t)(q * NT) + threadIdx.x;
            asm volatile("" : "+v"(ch));   // (per channel: nothing of this is to be kept in registers over the loop)
            const uint32_t row = sg.d_wzc.div(ch), cz = ch - row * (uint32_t)(WZ >> 2);
            const uint32_t lx = sg.d_wy.div(row), ly = row - lx * (uint32_t)WY;
            // (the look-up outside any divergent branch: ds_bpermute reads 0 from a lane that is switched off)
            const int2 o = org_of(min(cz >> 2, (uint32_t)sg.nseg - 1u));
            uint32_t src = DEAD;
            if (ch < sg.iw_chunks && cz * 4u < wezu)
                src = __umul24((uint32_t)clamp1(o.x + (int)lx, nx), gxB) + __umul24((uint32_t)clamp1(o.y + (int)ly, ny), gyB) +
                      ((uint32_t)wz0 + cz * 4u) * 4u;
            // the wave's LDS destination (M0) from a scalar made here: hoisted out of the channel loop it sits in a VGPR
            // for the whole loop (a spill at this kernel's 64)
            uint32_t wv = threadIdx.x >> 6;
            asm volatile("" : "+v"(wv));
            const uint32_t wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wv) * 64u;
            unsigned char *dst = lago_smem + iwB + ((size_t)(q * NT) + wbase) * 16u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rI, (__attribute__((address_space(3))) void *)dst, 16, src, 0, 0, 0);
        }
    };
    request_image(0);
    // how d_u starts: zero (the reference), the caller's d_u, or addgo * grad_out[component] -- ONE unconditional set of
    // loads (a branch around them would be joined with a wait for everything in flight): the planes are chosen by
    // wave-uniform selects, umode 0 reads nothing (offset beyond the plane: 0), 1 * x and 1 * 0 are exact
    {
        const float *sb = umode == 1 ? dun : gon;
        const float sm = umode == 2 ? addgo : 1.f;
        const BufRsrc r0 = make_rsrc(sb, planeB), r1 = make_rsrc(sb + nv, planeB), r2 = make_rsrc(sb + 2 * (size_t)nv, planeB);
#pragma unroll
        for (int it = 0; it < VPL; ++it) {
            const uint32_t off = umode ? SV[it] : DEAD;
            rux[it] = sm * buf_load1<float>(r0, off);
            ruy[it] = sm * buf_load1<float>(r1, off);
            ruz[it] = sm * buf_load1<float>(r2, off);
        }
    }

    // ---- per-voxel geometry, once (branch-free: dead voxels -- their u reads 0 -- compute along and end as NOWIN; the
    // origin look-ups need every lane of the wave switched on)
    uint32_t A0[VPL], A1[VPL];
    float FT[VPL], FU[VPL], FV[VPL];
#pragma unroll
    for (int it = 0; it < VPL; ++it) {
        const uint32_t tt = threadIdx.x + (uint32_t)it * NT;
        const uint32_t a = sg.d_TyTz.div(tt);
        const uint32_t rr = tt - a * (uint32_t)(sg.TY * sg.TZ);
        const uint32_t b = sg.d_Tz.div(rr);
        const uint32_t kk = rr - b * (uint32_t)sg.TZ;
        const float hx = shear_pos<UNIT>(x0 + (int)a, dt, ux[it]);
        const float hy = shear_pos<UNIT>(y0 + (int)b, dt, uy[it]);
        const float hz = shear_pos<UNIT>(z0 + (int)kk, dt, uz[it]);
        const int fx = lg_floor(hx), fy = lg_floor(hy), fz = lg_floor(hz);
        FT[it] = hx - (float)fx;
        FU[it] = hy - (float)fy;
        FV[it] = hz - (float)fz;
        // the two z cells, clamped as the reference clamps them, and the window segment each falls in
        const int cz0 = clamp1(fz, nz), cz1 = clamp1(fz + 1, nz);
        const uint32_t lz0 = (uint32_t)(cz0 - wz0), lz1 = (uint32_t)(cz1 - wz0);
        const uint32_t smax = (uint32_t)sg.nseg - 1u;
        const int2 o0 = org_of(min(lz0 >> 4, smax)), o1 = org_of(min(lz1 >> 4, smax));
        const uint32_t lx0 = (uint32_t)(fx - o0.x), ly0 = (uint32_t)(fy - o0.y);
        const uint32_t lx1 = (uint32_t)(fx - o1.x), ly1 = (uint32_t)(fy - o1.y);
        const bool inwin = SV[it] != DEAD && lz0 < wezu && lz1 < wezu && lx0 < wxu1 && ly0 < wyu1 && lx1 < wxu1 && ly1 < wyu1;
        A0[it] = inwin ? __umul24(lx0, sxB) + __umul24(ly0, syB) + lz0 * 8u : NOWIN;
        A1[it] = inwin ? __umul24(lx1, sxB) + __umul24(ly1, syB) + lz1 * 8u : NOWIN;
    }

    for (int c = 0; c < nc; ++c) {
        const float *Ic = In + (size_t)c * nv;
        const BufRsrc rdI = make_rsrc(dIn + (size_t)c * nv, planeB);
        float gvs[VPL];
#pragma unroll
        for (int it = 0; it < VPL; ++it) gvs[it] = pgv[it];
        // channel c's image window has landed (every wave waits for its own LDS-direct loads, then the barrier); the
        // accumulation window is zero (prologue / the previous flush)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (c + 1 < nc) {   // the next channel's grad_out values travel under this channel's work
            const BufRsrc rg = make_rsrc(gon + (size_t)(c + 1) * nv, planeB);
#pragma unroll
            for (int it = 0; it < VPL; ++it) pgv[it] = buf_load1<float>(rg, SV[it]);
        }
#pragma unroll
        for (int it = 0; it < VPL; ++it) {
            if (VPL > 1 && it) __builtin_amdgcn_sched_barrier(0);  // one pass after the other (register pressure)
            // the kept geometry is made opaque once per channel: otherwise everything derived from it is hoisted out
            // of the channel loop into registers (splat_shear_mc_kernel)
            asm volatile("" : "+v"(A0[it]), "+v"(A1[it]), "+v"(FT[it]), "+v"(FU[it]), "+v"(FV[it]), "+v"(SV[it]));
            const uint32_t sv = SV[it];
            const float gv = gvs[it];
            if (sv == DEAD) continue;
            const float t = FT[it], uu = FU[it], v = FV[it];
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
            const uint32_t a0 = A0[it], a1 = A1[it];
            float gx, gy, gz;
            if (a0 != NOWIN) {
                // corners in Lerp3's order: rows (fx,fy) (fx+1,fy) (fx+1,fy+1) (fx,fy+1) at the floor cell, then at the ceil cell
                const unsigned char *i0 = lago_smem + iwB + (a0 >> 1), *i1 = lago_smem + iwB + (a1 >> 1);
                const uint32_t sx4 = sxB >> 1, sy4 = syB >> 1;
                const float c0 = *reinterpret_cast<const float *>(i0), c4 = *reinterpret_cast<const float *>(i1);
                const float c1 = *reinterpret_cast<const float *>(i0 + sx4), c5 = *reinterpret_cast<const float *>(i1 + sx4);
                const float c2 = *reinterpret_cast<const float *>(i0 + sx4 + sy4), c6 = *reinterpret_cast<const float *>(i1 + sx4 + sy4);
                const float c3 = *reinterpret_cast<const float *>(i0 + sy4), c7 = *reinterpret_cast<const float *>(i1 + sy4);
                lds_add(reinterpret_cast<double *>(lago_smem + a0), (double)wq[0]);
                lds_add(reinterpret_cast<double *>(lago_smem + a1), (double)wq[1]);
                lds_add(reinterpret_cast<double *>(lago_smem + a0 + syB), (double)wq[2]);
                lds_add(reinterpret_cast<double *>(lago_smem + a1 + syB), (double)wq[3]);
                lds_add(reinterpret_cast<double *>(lago_smem + a0 + sxB), (double)wq[4]);
                lds_add(reinterpret_cast<double *>(lago_smem + a1 + sxB), (double)wq[5]);
                lds_add(reinterpret_cast<double *>(lago_smem + a0 + sxB + syB), (double)wq[6]);
                lds_add(reinterpret_cast<double *>(lago_smem + a1 + sxB + syB), (double)wq[7]);
                // include/interp.h:315-326
                gx = lg_fma(omv, lg_fma(omu, c1 - c0, uu * (c2 - c3)), v * lg_fma(omu, c5 - c4, uu * (c6 - c7)));
                gy = lg_fma(omv, lg_fma(omt, c3 - c0, t * (c2 - c1)), v * lg_fma(omt, c7 - c4, t * (c6 - c5)));
                gz = lg_fma(omu, lg_fma(omt, c4 - c0, t * (c5 - c1)), uu * lg_fma(omt, c7 - c3, t * (c6 - c2)));
            } else {
                // beyond the window: position again (same expressions, same bits), the reference's clamped global
                // atomics (include/interp.h:330-401, :431-453) and Lerp3's pair gathers
                uint32_t tt = threadIdx.x + (uint32_t)it * NT;
                asm volatile("" : "+v"(tt));  // nothing of this rare path may be hoisted out of the channel loop
                const uint32_t a = sg.d_TyTz.div(tt);
                const uint32_t rr = tt - a * (uint32_t)(sg.TY * sg.TZ);
                const uint32_t b = sg.d_Tz.div(rr);
                const uint32_t kk = rr - b * (uint32_t)sg.TZ;
                const float hx = shear_pos<UNIT>(x0 + (int)a, dt, un[sv >> 2]);
                const float hy = shear_pos<UNIT>(y0 + (int)b, dt, un[(sv >> 2) + nv]);
                const float hz = shear_pos<UNIT>(z0 + (int)kk, dt, un[(sv >> 2) + 2 * (size_t)nv]);
                const int fx = lg_floor(hx), fy = lg_floor(hy), fz = lg_floor(hz);
                const uint32_t X0 = __umul24((uint32_t)clamp1(fx, nx), gxB), X1 = __umul24((uint32_t)clamp1(fx + 1, nx), gxB);
                const uint32_t Y0 = __umul24((uint32_t)clamp1(fy, ny), gyB), Y1 = __umul24((uint32_t)clamp1(fy + 1, ny), gyB);
                const uint32_t Z0 = (uint32_t)clamp1(fz, nz) * 4u, Z1 = (uint32_t)clamp1(fz + 1, nz) * 4u;
                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[0], rdI, X0 + Y0 + Z0, 0, 0);
                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[1], rdI, X0 + Y0 + Z1, 0, 0);
                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[2], rdI, X0 + Y1 + Z0, 0, 0);
                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[3], rdI, X0 + Y1 + Z1, 0, 0);
                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[4], rdI, X1 + Y0 + Z0, 0, 0);
                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[5], rdI, X1 + Y0 + Z1, 0, 0);
                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[6], rdI, X1 + Y1 + Z0, 0, 0);
                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[7], rdI, X1 + Y1 + Z1, 0, 0);
                Lerp3<float, false> Lq;
                Lq.setup(hx, hy, hz, nx, ny, nz);
                Lq.grad(Ic, gx, gy, gz);
            }
            // cuda/interp.cu:230: (Real)((double)diff * dt); for dt = +-1 that is +-diff exactly
            const float diff = UNIT ? (float)dt * gv : (float)((double)gv * dt);
            rux[it] = lg_fma(gx, diff, rux[it]);   // ascending channel order, as the reference's thread-owned sum
            ruy[it] = lg_fma(gy, diff, ruy[it]);
            ruz[it] = lg_fma(gz, diff, ruz[it]);
        }
        if (c + 1 == nc) {   // d_u is complete: its stores travel under the flush
#pragma unroll
            for (int it = 0; it < VPL; ++it) {
                const uint32_t off = SV[it];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, rux[it]), make_rsrc(dun, planeB), off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, ruy[it]), make_rsrc(dun + nv, planeB), off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, ruz[it]), make_rsrc(dun + 2 * (size_t)nv, planeB), off, 0, 0);
            }
        } else {
            // the next channel's grad_out values are settled in front of the flush (loads and atomics share one in-order
            // counter: behind the flush the wait would also cover every flush atomic's acknowledgement)
#pragma unroll
            for (int it = 0; it < VPL; ++it) asm volatile("" : "+v"(pgv[it]));
        }
        __syncthreads();   // every add has landed, every corner has been read
        if (c + 1 < nc) request_image(c + 1);
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
                    const int2 o = org_of((uint32_t)lz >> 4);   // (source lanes < nseg are on whenever any lane is)
                    if (acc != 0.0) {
                        if (c + 1 < nc) wrow[lz] = 0.0;
                        const uint32_t off = __umul24((uint32_t)clamp1(o.x + (int)lx, nx), gxB) +
                                             __umul24((uint32_t)clamp1(o.y + (int)ly, ny), gyB) + (uint32_t)(wz0 + lz) * 4u;
                        (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32((float)acc, rdI, off, 0, 0);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Persistent, software-pipelined form of the image-window splat (round 4).
//
// tools/probes/atomic_overlap.hip: the C = 1 splat's traffic at 8 x 128^3 -- 537 MB streamed, 98 MB of flush atomics --
// takes 94 us and 74 us on their own and 123 us together however the work is arranged (float atomics execute at the
// memory side at 1.33 TB/s CHIP-wide: half the CUs flush at the same rate), so ~125 us is the floor of any
// window + atomic-flush scheme; the one-shot kernels above need 180-200 us because a barrier-phased workgroup exposes
// every memory round trip of its tile and only the CU's other workgroup covers it.  Here ONE 1024-thread workgroup per
// CU walks a sequence of (tile, channel) steps and the memory system is kept busy from inside the workgroup:
//   * everything step s + 1 needs is requested at the top of step s -- u and the d_u start values of a new tile and
//     grad_out of the step into registers, the step's image window by LDS-direct loads into the OTHER of two image
//     windows -- and is settled at the end of step s's compute phase, a whole phase later, IN FRONT of the d_u
//     stores and the flush atomics (loads, stores and atomics retire through one in-order counter: a wait placed
//     behind the flush would also wait ~1 us for the acknowledgement of every atomic);
//   * the segment origins of a new tile (the probe of the tile's centre column) are requested TWO steps ahead, so the
//     image-window addresses of step s + 1 never wait for a dependent load;
//   * the LDS-direct loads are issued from inline assembly: hipcc makes every LDS read behind a `buffer_load ... lds`
//     wait for vmcnt(0) (it cannot tell the image window from the accumulation window), which would serialise the
//     prefetch with the compute phase it is meant to hide under.  Their completion is waited for explicitly at the
//     settle point; the barrier behind it publishes the window.
// Per step: [barrier] requests for s + 1, corners + adds of s, settle, d_u stores, [barrier] flush of s (atomics,
// re-zero).  Geometry (window addresses, fractions) is computed once per tile and kept over its channels.  z rows are
// whole (nz <= 160, a multiple of 16: configs[1] / configs[4]); the float64 window plus two float32 image windows are
// 16 bytes per cell of the 160 KB LDS: tiles of 5 x 6 x 128 / 5 x 5 x 160 voxels.  Arithmetic per voxel and channel is
// that of splat_shear_kernel: d_u bit-identical; out-of-window samples as in splat_shear_iw_kernel.
#ifdef LAGO_PROFILING   // profiling builds only (results are WRONG with bits set): stages of the image-window kernels switched off
__device__ int g_dev_splat_skip = 0;   // 1 LDS adds, 2 corner reads + gradient, 4 image-window loads, 8 flush atomics,
                                       // 16 d_u stores, 32 voxel operand loads, 64 the whole flush
#define LAGO_SPLAT_SKIP(bit) (dbg_skip & (bit))
#define LAGO_SPLAT_SKIP_INIT const int dbg_skip = __builtin_amdgcn_readfirstlane(g_dev_splat_skip);
#else
#define LAGO_SPLAT_SKIP(bit) false
#define LAGO_SPLAT_SKIP_INIT
#endif
typedef unsigned int lg_rsrc4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ lg_rsrc4 raw_rsrc(const void *base, uint32_t bytes) {
    const unsigned long long a = (unsigned long long)base;
    lg_rsrc4 r = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
    return r;
}
// one LDS-direct 16-byte-per-lane load the compiler does not know to be one (see above); lds_base is wave-uniform
__device__ __forceinline__ void lds_dma16(lg_rsrc4 r, uint32_t lds_base, uint32_t voff) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_base), "v"(voff), "s"(r) : "memory", "m0");
#pragma clang diagnostic pop
}

template <bool UNIT, bool BC, int VPL, bool MC, int PARTS>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void splat_shear_pp_kernel(
    float *__restrict__ d_I, float *__restrict__ d_u, const float *__restrict__ go, const float *__restrict__ I,
    const float *__restrict__ u, double dt, int nc, ShearGeom sg, int umode, float addgo) {
    constexpr int NT = 1024;
    constexpr uint32_t NOWIN = 0xffffffffu, DEAD = 0xffffffffu;
    LAGO_SPLAT_SKIP_INIT
    constexpr int RCH = 3;    // image-window chunks per thread (host: iw_chunks <= RCH * NT)
    // window cells a thread flushes per step: FROWS window rows per wave (host: WX * WY <= 16 * FROWS) x PARTS 64-cell parts of a row
    constexpr int FROWS = PARTS <= 2 ? 6 : 4, NFL = FROWS * PARTS;
    extern __shared__ __align__(16) unsigned char lago_smem[];
    double *win = reinterpret_cast<double *>(lago_smem);
    const uint32_t iwB0 = sg.win_cells * 8u;   // byte offset of image window 0; window 1 follows it
    const int nx = sg.nx, ny = sg.ny, nz = sg.nz;
    const uint32_t nv = (uint32_t)nx * ny * nz;
    const uint32_t planeB = nv * 4u;
    const int WX = sg.WX, WY = sg.WY, WZ = sg.WZ;   // WZ == nz: whole z rows, window z origin 0
    const uint32_t sxB = (uint32_t)(WY * WZ) * 8u, syB = (uint32_t)WZ * 8u;            // window strides in bytes
    const uint32_t gxB = (uint32_t)ny * nz * 4u, gyB = (uint32_t)nz * 4u;              // grid strides in bytes
    const uint32_t wxu1 = (uint32_t)(WX - 1), wyu1 = (uint32_t)(WY - 1);
    const uint32_t smax = (uint32_t)sg.nseg - 1u;
    const int lane = (int)(threadIdx.x & 63u);
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));

    // ---- this workgroup's tiles: XCD x (workgroups w = x mod 8) walks the x-th eighth of the launch, all its workgroups
    // side by side (neighbouring tiles are in flight together: halo rows are L2 hits); the < 8 tiles left over go to the
    // first workgroups as one more tile each
    const uint32_t G = gridDim.x, w = blockIdx.x, q = sg.total >> 3, g8 = G >> 3, rem = sg.total - (q << 3);
    const uint32_t nmain = q > (w >> 3) ? (q - (w >> 3) + g8 - 1) / g8 : 0u;
    const uint32_t ntiles = nmain + (w < rem ? 1u : 0u);
    struct Tile {
        uint32_t n;
        int x0, y0, ex, ey;
    };
    auto tile_at = [&](uint32_t k) {
        uint32_t L = k < nmain ? (w & 7u) * q + (w >> 3) + k * g8 : (q << 3) + w;
        if (sg.rev) L = sg.total - 1u - L;
        Tile t;
        t.n = sg.d_tiles.div(L);
        const uint32_t r = L - t.n * sg.tiles_per_item;
        const uint32_t bx = sg.d_tyz.div(r), by = r - bx * sg.nty;   // (ntz == 1)
        t.x0 = (int)bx * sg.TX;
        t.y0 = (int)by * sg.TY;
        t.ex = min(sg.TX, nx - t.x0);
        t.ey = min(sg.TY, ny - t.y0);
        return t;
    };
    // ---- what is the same for every tile, once: the thread's voxels (a | b << 8 | z << 16 within the tile; a = 255:
    // none) and its image-window chunks (lx | ly << 8 | z chunk << 16; all ones: none).  Decoding a voxel per use costs
    // six quarter-rate 32-bit multiplies (FastDiv), three uses per step.
    uint32_t vk[VPL], ck[RCH];
#pragma unroll
    for (int it = 0; it < VPL; ++it) {
        const uint32_t tt = threadIdx.x + (uint32_t)it * NT;
        const uint32_t a = sg.d_TyTz.div(tt);
        const uint32_t rr = tt - a * (uint32_t)(sg.TY * sg.TZ);
        const uint32_t b = sg.d_Tz.div(rr);
        const uint32_t kk = rr - b * (uint32_t)sg.TZ;
        vk[it] = tt < sg.tile_vox ? a | (b << 8) | (kk << 16) : 255u;
    }
#pragma unroll
    for (int qq = 0; qq < RCH; ++qq) {
        const uint32_t ch = (uint32_t)(qq * NT) + threadIdx.x;
        const uint32_t row = sg.d_wzc.div(ch), cz = ch - row * (uint32_t)(WZ >> 2);
        const uint32_t lx = sg.d_wy.div(row), ly = row - lx * (uint32_t)WY;
        ck[qq] = ch < sg.iw_chunks ? lx | (ly << 8) | (cz << 16) : DEAD;
    }
    // segment origins: lane s < nseg of EVERY wave holds the (x, y) origin of z segment s, packed (x + 1) << 16 | (y + 1);
    // a look-up is one ds_bpermute (splat_shear_iw_kernel)
    auto probe_issue = [&](const Tile &t, float &pox, float &poy) {
        const float *un = u + (size_t)t.n * 3 * nv;
        const int cxs = t.x0 + t.ex / 2, cys = t.y0 + t.ey / 2;
        const int zc = min(lane * 16 + 8, nz - 1);
        const uint32_t off = lane < sg.nseg ? (((uint32_t)cxs * ny + cys) * nz + zc) * 4u : DEAD;
        pox = buf_load1<float>(make_rsrc(un, planeB), off);
        poy = buf_load1<float>(make_rsrc(un + nv, planeB), off);
    };
    auto org_pack = [&](const Tile &t, float pox, float poy) {
        const float fdt = (float)dt;
        const int ox = max(-1, min(t.x0 + (int)floorf(fdt * pox) - sg.MX, nx + 1 - WX));
        const int oy = max(-1, min(t.y0 + (int)floorf(fdt * poy) - sg.MY, ny + 1 - WY));
        return ((ox + 1) << 16) | (oy + 1);
    };
    auto org_of = [&](int orgp, uint32_t seg) {
        const int p = __builtin_amdgcn_ds_bpermute((int)(seg << 2), orgp);
        int2 o;
        o.x = (p >> 16) - 1;
        o.y = (p & 0xffff) - 1;
        return o;
    };
    // voxel it of this thread in tile t: byte offset in a plane, or DEAD
    auto voxel_off = [&](const Tile &t, int it) {
        const uint32_t a = vk[it] & 255u, b = (vk[it] >> 8) & 255u, kk = vk[it] >> 16;
        const bool live = (int)a < t.ex && (int)b < t.ey;
        const uint32_t row = __umul24((uint32_t)t.x0 + a, (uint32_t)ny) + (uint32_t)t.y0 + b;   // (host: nx * ny < 2^24)
        return live ? (row * (uint32_t)nz + kk) * 4u : DEAD;
    };
    float Pux[VPL], Puy[VPL], Puz[VPL], Pgv[VPL];
    float Psx[MC ? VPL : 1], Psy[MC ? VPL : 1], Psz[MC ? VPL : 1];   // d_u start values of the NEXT tile (umode != 0)
    // u (+ d_u start values) and grad_out of voxel `it` of tile t (a new tile's operands)
    auto voxel_issue = [&](const Tile &t, int it, bool with_u, int c) {
        const uint32_t off = LAGO_SPLAT_SKIP(32) ? DEAD : voxel_off(t, it);
        if (with_u) {
            const float *un = u + (size_t)t.n * 3 * nv;
            Pux[it] = buf_load1<float>(make_rsrc(un, planeB), off);
            Puy[it] = buf_load1<float>(make_rsrc(un + nv, planeB), off);
            Puz[it] = buf_load1<float>(make_rsrc(un + 2 * (size_t)nv, planeB), off);
            if (MC) {   // (ONE unconditional set of loads: planes by wave-uniform selects, umode 0 reads nothing)
                const float *sb = umode == 1 ? d_u + (size_t)t.n * 3 * nv : go + (size_t)t.n * nc * nv;
                const uint32_t so = umode ? off : DEAD;
                Psx[it] = buf_load1<float>(make_rsrc(sb, planeB), so);
                Psy[it] = buf_load1<float>(make_rsrc(sb + nv, planeB), so);
                Psz[it] = buf_load1<float>(make_rsrc(sb + 2 * (size_t)nv, planeB), so);
            }
        }
        Pgv[it] = buf_load1<float>(make_rsrc(go + ((size_t)t.n * nc + c) * nv, planeB), off);
    };
    // chunk qq of the image window of (tile t, channel c) -> image window `buf`: four z cells from the clamped grid row
    // of their segment's origin
    auto image_issue = [&](const Tile &t, int c, int orgp, uint32_t buf, int qq) {
        if ((uint32_t)(qq * NT) + wave * 64u >= sg.iw_chunks) return;   // wave-uniform
        const lg_rsrc4 rI = raw_rsrc(I + ((BC ? (size_t)0 : (size_t)t.n * nc) + c) * nv, planeB);
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lago_smem + iwB0 + buf * sg.iw_bytes;
        const uint32_t lx = ck[qq] & 255u, ly = (ck[qq] >> 8) & 255u, cz = (ck[qq] >> 16) & 255u;
        const int2 o = org_of(orgp, min(cz >> 2, smax));
        const uint32_t src = ck[qq] != DEAD ? __umul24((uint32_t)clamp1(o.x + (int)lx, nx), gxB) +
                                                  __umul24((uint32_t)clamp1(o.y + (int)ly, ny), gyB) + cz * 16u
                                            : DEAD;
        if (!LAGO_SPLAT_SKIP(4)) lds_dma16(rI, lds0 + ((uint32_t)(qq * NT) + wave * 64u) * 16u, src);
    };

    if (ntiles == 0) return;
    // ---- prologue: what the top of step 0 expects to be there
    Tile Tc = tile_at(0), Tn = Tc;   // current tile; tile of the next step
    int orgc, orgn = 0;              // their packed origins
    float pox, poy;                  // probe in flight (tile of step s + 2, when that step starts a tile)
    {
        probe_issue(Tc, pox, poy);
#pragma unroll
        for (int it = 0; it < VPL; ++it) voxel_issue(Tc, it, true, 0);
        double2 *w2 = reinterpret_cast<double2 *>(win);
        for (uint32_t f = threadIdx.x; f < sg.win_cells / 2; f += NT) w2[f] = make_double2(0.0, 0.0);
        orgc = org_pack(Tc, pox, poy);
#pragma unroll
        for (int qq = 0; qq < RCH; ++qq) image_issue(Tc, 0, orgc, 0u, qq);
        if (nc == 1 && ntiles > 1) probe_issue(tile_at(1), pox, poy);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    uint32_t SV[VPL], A0[VPL], A1[VPL];
    float FT[VPL], FU[VPL], FV[VPL];
    float rux[MC ? VPL : 1], ruy[MC ? VPL : 1], ruz[MC ? VPL : 1];
    // the previous step's flush, captured: the values of this thread's cells (rounded once, as the flush always did; 0 =
    // untouched), the d_I plane (in elements) they go to and the origins of the tile they belong to
    float fval[NFL];
#pragma unroll
    for (int f = 0; f < NFL; ++f) fval[f] = 0.f;
    size_t fplane = 0;
    int orgf = 0;
    // cells f0 .. f1 - 1 of it: cell f = (window row wave + 16 (f / PARTS), z = lane + 64 (f % PARTS))
    auto flush_emit = [&](const BufRsrc &rfl, int f0, int f1) {
        int2 so[PARTS];
#pragma unroll
        for (int j = 0; j < PARTS; ++j) so[j] = org_of(orgf, min((uint32_t)(lane + 64 * j) >> 4, smax));
#pragma unroll
        for (int f = 0; f < NFL; ++f) {
            if (f < f0 || f >= f1) continue;
            const uint32_t row = wave + (uint32_t)(f / PARTS) * (NT / 64);
            const uint32_t lx = sg.d_wy.div(row), ly = row - lx * (uint32_t)WY;   // (scalar)
            const int j = f % PARTS;
            const uint32_t off = __umul24((uint32_t)clamp1(so[j].x + (int)lx, nx), gxB) +
                                 __umul24((uint32_t)clamp1(so[j].y + (int)ly, ny), gyB) + (uint32_t)(lane + 64 * j) * 4u;
            if (!LAGO_SPLAT_SKIP(8))
                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(fval[f], rfl, fval[f] != 0.f ? off : 0x80000000u, 0, 0);   // (beyond any plane, no 32-bit wrap)
        }
    };
    uint32_t k = 0, buf = 0;
    int c = 0;
    bool more = true;
    __syncthreads();   // the accumulation window is zero; image window 0 is published
    while (more) {
        const bool first = c == 0;
        const bool next_same = c + 1 < nc;                  // step s + 1: the same tile
        const bool have_next = next_same || k + 1 < ntiles;
        if (first) {
            // ---- geometry of the tile, once (branch-free: dead voxels -- their u reads 0 -- compute along and end as
            // NOWIN; the origin look-ups need every lane of the wave switched on)
#pragma unroll
            for (int it = 0; it < VPL; ++it) {
                const uint32_t a = vk[it] & 255u, b = (vk[it] >> 8) & 255u, kk = vk[it] >> 16;
                SV[it] = voxel_off(Tc, it);
                const float hx = shear_pos<UNIT>(Tc.x0 + (int)a, dt, Pux[it]);
                const float hy = shear_pos<UNIT>(Tc.y0 + (int)b, dt, Puy[it]);
                const float hz = shear_pos<UNIT>((int)kk, dt, Puz[it]);
                const int fx = lg_floor(hx), fy = lg_floor(hy), fz = lg_floor(hz);
                FT[it] = hx - (float)fx;
                FU[it] = hy - (float)fy;
                FV[it] = hz - (float)fz;
                const uint32_t lz0 = (uint32_t)clamp1(fz, nz), lz1 = (uint32_t)clamp1(fz + 1, nz);   // always inside the window
                const int2 o0 = org_of(orgc, lz0 >> 4), o1 = org_of(orgc, lz1 >> 4);
                const uint32_t lx0 = (uint32_t)(fx - o0.x), ly0 = (uint32_t)(fy - o0.y);
                const uint32_t lx1 = (uint32_t)(fx - o1.x), ly1 = (uint32_t)(fy - o1.y);
                const bool inwin = SV[it] != DEAD && lx0 < wxu1 && ly0 < wyu1 && lx1 < wxu1 && ly1 < wyu1;
                A0[it] = inwin ? __umul24(lx0, sxB) + __umul24(ly0, syB) + lz0 * 8u : NOWIN;
                A1[it] = inwin ? __umul24(lx1, sxB) + __umul24(ly1, syB) + lz1 * 8u : NOWIN;
                if (MC) {
                    const float sm = umode == 2 ? addgo : 1.f;   // (1 * x and 1 * 0 are exact)
                    rux[it] = sm * Psx[it]; ruy[it] = sm * Psy[it]; ruz[it] = sm * Psz[it];
                }
            }
        }
        float gvs[VPL];
#pragma unroll
        for (int it = 0; it < VPL; ++it) gvs[it] = Pgv[it];
        // ---- what step s + 1 will need (requested pass by pass below), and the probe of the tile that starts at s + 2
        if (have_next && !next_same) {
            Tn = tile_at(k + 1);
            orgn = org_pack(Tn, pox, poy);   // (its probe was settled at the end of the previous step)
        }
        if (have_next) {
            if (nc == 1) {
                if (k + 2 < ntiles) probe_issue(tile_at(k + 2), pox, poy);
            } else if (c + 2 == nc && k + 1 < ntiles) {
                probe_issue(tile_at(k + 1), pox, poy);
            }
        }
        const Tile &Tq = next_same ? Tc : Tn;             // tile, channel and origin of step s + 1
        const int cq = next_same ? c + 1 : 0;
        const int orgq = next_same ? orgc : orgn;
        // ---- corners, adds, gradient of step s; between the passes the requests of step s + 1 and the atomics of step
        // s - 1's flush, a few at a time: loads and atomics share the CU's in-order memory pipeline, and a burst of
        // either kind holds the other up (tools/probes/atomic_overlap.hip: 94 + 74 us as bursts, 123 us interleaved)
        const float *Ic = I + ((BC ? (size_t)0 : (size_t)Tc.n * nc) + c) * nv;
        const size_t cplane = ((BC ? (size_t)0 : (size_t)Tc.n * nc) + c);
        const BufRsrc rdI = make_rsrc(d_I + cplane * nv, planeB);
        const BufRsrc rfl = make_rsrc(d_I + fplane * nv, planeB);
        const uint32_t iwB = iwB0 + buf * sg.iw_bytes;
        float *dun = d_u + (size_t)Tc.n * 3 * nv;
        const bool last = c + 1 == nc;
        float ox_[VPL], oy_[VPL], oz_[VPL];   // d_u of the tile's voxels (stored behind the settle point)
#pragma unroll
        for (int it = 0; it < VPL; ++it) {
            if (VPL > 1 && it) __builtin_amdgcn_sched_barrier(0);  // one pass after the other (register pressure)
            asm volatile("" : "+v"(A0[it]), "+v"(A1[it]), "+v"(FT[it]), "+v"(FU[it]), "+v"(FV[it]), "+v"(SV[it]));
            ox_[it] = oy_[it] = oz_[it] = 0.f;
            const uint32_t sv = SV[it];
            const float gv = gvs[it];
            if (sv != DEAD) {
                const float t = FT[it], uu = FU[it], v = FV[it];
                const float omt = 1.f - t, omu = 1.f - uu, omv = 1.f - v;
                // sequentially flipped weights (include/interp.h:431-453): x outer, y, z inner
                float wq[8];
                {
                    float ddx = omt, ddy = omu, ddz = omv;
#pragma unroll
                    for (int qq = 0; qq < 8; ++qq) {
                        wq[qq] = (ddx * ddy * ddz) * gv;
                        ddz = 1.f - ddz;
                        if (qq & 1) ddy = 1.f - ddy;
                        if ((qq & 3) == 3) ddx = 1.f - ddx;
                    }
                }
                const uint32_t a0 = A0[it], a1 = A1[it];
                float gx, gy, gz;
                if (a0 != NOWIN) {
                    // corners in Lerp3's order: rows (fx,fy) (fx+1,fy) (fx+1,fy+1) (fx,fy+1) at the floor cell, then at the ceil cell
                    const unsigned char *i0 = lago_smem + iwB + (a0 >> 1), *i1 = lago_smem + iwB + (a1 >> 1);
                    const uint32_t sx4 = sxB >> 1, sy4 = syB >> 1;
                    float c0 = wq[0], c1 = wq[1], c2 = wq[2], c3 = wq[3], c4 = wq[4], c5 = wq[5], c6 = wq[6], c7 = wq[7];   // (profiling stand-ins)
                    if (!LAGO_SPLAT_SKIP(2)) {
                        c0 = *reinterpret_cast<const float *>(i0); c4 = *reinterpret_cast<const float *>(i1);
                        c1 = *reinterpret_cast<const float *>(i0 + sx4); c5 = *reinterpret_cast<const float *>(i1 + sx4);
                        c2 = *reinterpret_cast<const float *>(i0 + sx4 + sy4); c6 = *reinterpret_cast<const float *>(i1 + sx4 + sy4);
                        c3 = *reinterpret_cast<const float *>(i0 + sy4); c7 = *reinterpret_cast<const float *>(i1 + sy4);
                    }
                    if (!LAGO_SPLAT_SKIP(1)) {
                        lds_add(reinterpret_cast<double *>(lago_smem + a0), (double)wq[0]);
                        lds_add(reinterpret_cast<double *>(lago_smem + a1), (double)wq[1]);
                        lds_add(reinterpret_cast<double *>(lago_smem + a0 + syB), (double)wq[2]);
                        lds_add(reinterpret_cast<double *>(lago_smem + a1 + syB), (double)wq[3]);
                        lds_add(reinterpret_cast<double *>(lago_smem + a0 + sxB), (double)wq[4]);
                        lds_add(reinterpret_cast<double *>(lago_smem + a1 + sxB), (double)wq[5]);
                        lds_add(reinterpret_cast<double *>(lago_smem + a0 + sxB + syB), (double)wq[6]);
                        lds_add(reinterpret_cast<double *>(lago_smem + a1 + sxB + syB), (double)wq[7]);
                    }
                    // include/interp.h:315-326
                    gx = lg_fma(omv, lg_fma(omu, c1 - c0, uu * (c2 - c3)), v * lg_fma(omu, c5 - c4, uu * (c6 - c7)));
                    gy = lg_fma(omv, lg_fma(omt, c3 - c0, t * (c2 - c1)), v * lg_fma(omt, c7 - c4, t * (c6 - c5)));
                    gz = lg_fma(omu, lg_fma(omt, c4 - c0, t * (c5 - c1)), uu * lg_fma(omt, c7 - c3, t * (c6 - c2)));
                } else {
                    // beyond the window: position again (same expressions, same bits), the reference's clamped global
                    // atomics (include/interp.h:330-401, :431-453) and Lerp3's pair gathers
                    uint32_t vkk = vk[it];
                    asm volatile("" : "+v"(vkk));  // nothing of this rare path may be hoisted out of the step loop
                    const uint32_t a = vkk & 255u, b = (vkk >> 8) & 255u, kk = vkk >> 16;
                    const float *un = u + (size_t)Tc.n * 3 * nv;
                    const float hx = shear_pos<UNIT>(Tc.x0 + (int)a, dt, un[sv >> 2]);
                    const float hy = shear_pos<UNIT>(Tc.y0 + (int)b, dt, un[(sv >> 2) + nv]);
                    const float hz = shear_pos<UNIT>((int)kk, dt, un[(sv >> 2) + 2 * (size_t)nv]);
                    const int fx = lg_floor(hx), fy = lg_floor(hy), fz = lg_floor(hz);
                    const uint32_t X0 = __umul24((uint32_t)clamp1(fx, nx), gxB), X1 = __umul24((uint32_t)clamp1(fx + 1, nx), gxB);
                    const uint32_t Y0 = __umul24((uint32_t)clamp1(fy, ny), gyB), Y1 = __umul24((uint32_t)clamp1(fy + 1, ny), gyB);
                    const uint32_t Z0 = (uint32_t)clamp1(fz, nz) * 4u, Z1 = (uint32_t)clamp1(fz + 1, nz) * 4u;
                    (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[0], rdI, X0 + Y0 + Z0, 0, 0);
                    (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[1], rdI, X0 + Y0 + Z1, 0, 0);
                    (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[2], rdI, X0 + Y1 + Z0, 0, 0);
                    (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[3], rdI, X0 + Y1 + Z1, 0, 0);
                    (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[4], rdI, X1 + Y0 + Z0, 0, 0);
                    (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[5], rdI, X1 + Y0 + Z1, 0, 0);
                    (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[6], rdI, X1 + Y1 + Z0, 0, 0);
                    (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wq[7], rdI, X1 + Y1 + Z1, 0, 0);
                    Lerp3<float, false> Lq;
                    Lq.setup(hx, hy, hz, nx, ny, nz);
                    Lq.grad(Ic, gx, gy, gz);
                }
                // cuda/interp.cu:230: (Real)((double)diff * dt); for dt = +-1 that is +-diff exactly
                const float diff = UNIT ? (float)dt * gv : (float)((double)gv * dt);
                // ascending channel order, as the reference's thread-owned sum
                if (MC) {
                    rux[it] = lg_fma(gx, diff, rux[it]);
                    ruy[it] = lg_fma(gy, diff, ruy[it]);
                    ruz[it] = lg_fma(gz, diff, ruz[it]);
                    ox_[it] = rux[it]; oy_[it] = ruy[it]; oz_[it] = ruz[it];
                } else {
                    ox_[it] = lg_fma(gx, diff, 0.f);
                    oy_[it] = lg_fma(gy, diff, 0.f);
                    oz_[it] = lg_fma(gz, diff, 0.f);
                }
            }
            // -- this pass's share of the requests for step s + 1 and of the previous step's flush (none in the last
            // pass: the settle below waits for the YOUNGEST operation too; a load returns ~1-2 us after it was issued,
            // an atomic is acknowledged ~1 us after)
            if (VPL == 1 || it + 1 < VPL) {
                constexpr int NP = VPL == 1 ? 1 : VPL - 1;   // passes that carry requests
                if (have_next) {
#pragma unroll
                    for (int qq = 0; qq < RCH; ++qq)
                        if (qq * NP / RCH == it) image_issue(Tq, cq, orgq, buf ^ 1u, qq);
#pragma unroll
                    for (int j = 0; j < VPL; ++j)
                        if (j * NP / VPL == it) voxel_issue(Tq, j, !next_same, cq);
                }
                constexpr int per = (NFL + NP - 1) / NP;
                flush_emit(rfl, it * per, (it + 1) * per);
            }
        }
        // ---- settle what was requested above: this wave's LDS-direct loads (invisible to the compiler) and, by naming
        // them, every prefetched register -- IN FRONT of the d_u stores below
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < VPL; ++it) {
            asm volatile("" : "+v"(Pux[it]), "+v"(Puy[it]), "+v"(Puz[it]), "+v"(Pgv[it]));
            if (MC) asm volatile("" : "+v"(Psx[it]), "+v"(Psy[it]), "+v"(Psz[it]));
        }
        asm volatile("" : "+v"(pox), "+v"(poy));
        if (last && !LAGO_SPLAT_SKIP(16)) {
#pragma unroll
            for (int it = 0; it < VPL; ++it) {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, ox_[it]), make_rsrc(dun, planeB), SV[it], 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, oy_[it]), make_rsrc(dun + nv, planeB), SV[it], 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, oz_[it]), make_rsrc(dun + 2 * (size_t)nv, planeB), SV[it], 0, 0);
            }
        }
        __syncthreads();   // every add has landed; the next step's image window is complete
        // ---- capture the flush of step s: this thread's touched cells (one wave per window row, lanes along z) leave the
        // window -- value and target offset into registers, the cell re-zeroed; the atomics follow during step s + 1
        if (!LAGO_SPLAT_SKIP(64)) {
            const uint32_t nrows = (uint32_t)(WX * WY);
#pragma unroll
            for (int rr = 0; rr < FROWS; ++rr) {
                const uint32_t row = wave + (uint32_t)rr * (NT / 64);
#pragma unroll
                for (int j = 0; j < PARTS; ++j) {
                    const int lz = lane + 64 * j;
                    float val = 0.f;
                    if (row < nrows && lz < WZ) {
                        double *cell = win + row * (uint32_t)WZ + lz;
                        const double acc = *cell;
                        if (acc != 0.0) {
                            *cell = 0.0;
                            val = (float)acc;
                        }
                    }
                    fval[rr * PARTS + j] = val;
                }
            }
        }
        orgf = orgc;
        fplane = cplane;
        more = have_next;
        if (next_same) ++c;
        else {
            c = 0;
            ++k;
            Tc = Tn;
            orgc = orgn;
        }
        buf ^= 1u;
        __syncthreads();   // the window is zero again before the next step adds to it
    }
    // ---- the last step's flush
    {
        const BufRsrc rfl = make_rsrc(d_I + fplane * nv, planeB);
        flush_emit(rfl, 0, NFL);
    }
}

// TX TY TZ(0 = auto) margins MX MY MZ.  TX is an upper bound: make_shear shrinks it until the float64 window fits
// 80 KB (8 x 6 -> 5 x 6 x 128 at nz = 128, 8 x 6 x 80 at nz = 160: 3840-voxel tiles).  Measured against 4 x 8
// (tools/ab_tiles.py, steady state): 1-3 % faster at 128^3 and 160^3, one and three channels.
static KnobArray<6> g_shear_cfg({8, 6, 0, 1, 1, 4});
// g_shear_mc, several channels with d_u wanted: 0 d_u read-modify-written per channel, 1 d_u in registers over the
// channels (splat_shear_kernel<..., VPL>), 2 (default) the geometry-once kernel (splat_shear_mc_kernel)
static std::atomic<int> g_shear_nt{1024}, g_shear_on{1}, g_shear_mc{2};
// 1 (default): d_u wanted -> splat_shear_iw_kernel (corners of the d_u term from an LDS window of I)
static std::atomic<int> g_shear_iw{1};
// 1: d_u wanted and whole z rows of at most 160 voxels -> splat_shear_pp_kernel (persistent, software-pipelined)
static std::atomic<int> g_shear_pp{1};
// 1: one channel (or no d_u) -> splat_row_kernel (index arithmetic out of the voxel loop)
static std::atomic<int> g_shear_row{1};

// max_vox > 0 (the geometry-once multi-channel kernel): the tile is shrunk further, larger of TX / TY first, until it
// has at most that many voxels.
// image_window (splat_shear_iw_kernel): 12 bytes per cell -- the float64 window plus the same cells of I as float32 --
// within `budget` bytes of LDS per workgroup (80 KB: two workgroups per CU).
static bool make_shear(ShearGeom &sg, const Geom &g, int64_t nn, size_t &smem, int max_vox = 0, bool image_window = false,
                       size_t budget = 80 * 1024) {
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
        const uint64_t cells = (uint64_t)sg.WX * sg.WY * sg.WZ;
        const uint64_t iwb = image_window ? ((cells / 4 + 63) / 64) * 1024 : 0;   // whole 1 KB wave-instructions
        const bool fits = cells * 8 + iwb + (uint64_t)sg.nseg * 8 <= budget;
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
    sg.iw_chunks = sg.iw_bytes = 0;
    if (image_window) {
        if ((sg.WZ & 15) || (g.nz & 3)) return false;   // 16-byte chunks of whole segments, rows 16-byte aligned
        sg.iw_chunks = sg.win_cells / 4;
        sg.iw_bytes = ((sg.iw_chunks + 63) / 64) * 1024;
        sg.d_wzc = FastDiv((uint32_t)(sg.WZ / 4));
    }
    smem = (size_t)sg.win_cells * sizeof(double) + sg.iw_bytes + (size_t)sg.nseg * 8;
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

// Geometry of splat_shear_pp_kernel: whole z rows (WZ = nz, a multiple of 16, at most 160), the float64 window plus two
// float32 image windows (16 bytes per cell) within the CU's 160 KB, tiles of at most 4096 voxels (four per thread).
static bool make_shear_pp(ShearGeom &sg, const Geom &g, int64_t nn, size_t &smem, int max_vox) {
    const std::array<int, 6> cfg = g_shear_cfg.get();
    int TX = cfg[0], TY = cfg[1];
    const int EX = cfg[3], EY = cfg[4];
    if (g.nz < 16 || (g.nz & 15) || g.nz > 160 || TX < 1 || TY < 1 || EX < 0 || EY < 0) return false;
    if ((uint64_t)g.ny * g.nz * 4 >= (1u << 24) || g.nx >= 32768 || g.ny >= 32768) return false;
    TX = TX < g.nx ? TX : g.nx;
    TY = TY < g.ny ? TY : g.ny;
    for (;;) {
        sg.WX = TX + 1 + 2 * EX;
        sg.WY = TY + 1 + 2 * EY;
        const uint64_t cells = (uint64_t)sg.WX * sg.WY * g.nz;
        const int frows = g.nz <= 128 ? 6 : 4;   // window rows a wave flushes per step (splat_shear_pp_kernel: FROWS)
        if (cells * 16 <= 160 * 1024 && (int64_t)TX * TY * g.nz <= max_vox && sg.WX * sg.WY <= 16 * frows) break;
        if (TX >= TY && TX > 1) --TX;
        else if (TY > 1) --TY;
        else return false;
    }
    if ((uint64_t)sg.WY * g.nz * 8 >= (1u << 24) || (int64_t)TX * TY * g.nz < 512) return false;
    sg.nx = g.nx; sg.ny = g.ny; sg.nz = g.nz;
    sg.TX = TX; sg.TY = TY; sg.TZ = g.nz;
    sg.WZ = g.nz;
    sg.MX = EX; sg.MY = EY; sg.MZ = 0;
    sg.nseg = g.nz / 16;
    sg.win_cells = (uint32_t)sg.WX * sg.WY * sg.WZ;
    sg.iw_chunks = sg.win_cells / 4;
    sg.iw_bytes = ((sg.iw_chunks + 63) / 64) * 1024;
    if (sg.iw_chunks > 3u * 1024u) return false;
    smem = (size_t)sg.win_cells * sizeof(double) + 2 * (size_t)sg.iw_bytes;
    if (smem > 160 * 1024) return false;
    sg.ntx = (g.nx + TX - 1) / TX;
    sg.nty = (g.ny + TY - 1) / TY;
    sg.ntz = 1;
    sg.tiles_per_item = sg.ntx * sg.nty;
    const int64_t total = (int64_t)sg.tiles_per_item * nn;
    if (total < 8 || total >= (1ll << 31)) return false;
    sg.total = (uint32_t)total;
    sg.rev = g.rev;
    sg.tile_vox = (uint32_t)TX * TY * sg.TZ;
    sg.d_tiles = FastDiv(sg.tiles_per_item);
    sg.d_tyz = FastDiv(sg.nty);
    sg.d_tz = FastDiv(1u);
    sg.d_TyTz = FastDiv((uint32_t)(TY * sg.TZ));
    sg.d_Tz = FastDiv((uint32_t)sg.TZ);
    sg.d_wy = FastDiv((uint32_t)sg.WY);
    sg.d_wzc = FastDiv((uint32_t)(sg.WZ / 4));
    return true;
}

// Geometry of splat_row_kernel: TY x TZ threads (a multiple of 64, at most 1024), TX passes; TY from the tile setting when
// that gives whole waves, else as many rows as 1024 threads hold.
static bool make_shear_row(ShearGeom &sg, const Geom &g, int64_t nn, size_t &smem, int &nthreads) {
    const std::array<int, 6> cfg = g_shear_cfg.get();
    int TX = cfg[0], TY = cfg[1], TZ = cfg[2];
    const int EX = cfg[3], EY = cfg[4], EZ = cfg[5];
    if (g.nz < 16 || TX < 1 || TY < 1 || EX < 0 || EY < 0 || EZ < 0) return false;
    if ((uint64_t)g.ny * g.nz * 4 >= (1u << 24) || g.nx >= (1 << 23)) return false;
    if (TZ <= 0) {  // auto: whole z rows up to 128 voxels, else even parts of at most 128 (multiples of 16)
        const int parts = (g.nz + 127) / 128;
        TZ = (((g.nz + parts - 1) / parts + 15) / 16) * 16;
    }
    TZ = TZ < g.nz ? TZ : g.nz;
    if (TZ > 1024) return false;
    if (TY > g.ny) TY = g.ny;
    if ((TY * TZ) % 64 != 0 || TY * TZ > 1024) {
        TY = 1024 / TZ;
        if (TY > g.ny) TY = g.ny;
        while (TY > 1 && (TY * TZ) % 64 != 0) --TY;
        if ((TY * TZ) % 64 != 0) return false;
    }
    TX = TX < g.nx ? TX : g.nx;
    for (;;) {
        sg.WX = TX + 1 + 2 * EX;
        sg.WY = TY + 1 + 2 * EY;
        sg.WZ = TZ >= g.nz ? g.nz : ((TZ + 1 + 2 * EZ + 15 + 15) / 16) * 16;
        if (sg.WZ > g.nz) sg.WZ = g.nz;
        sg.nseg = (sg.WZ + 15) / 16;
        if ((uint64_t)sg.WX * sg.WY * sg.WZ * 8 + (uint64_t)sg.nseg * 8 <= 80 * 1024) break;  // two workgroups per CU
        if (TX > 1) --TX;
        else return false;
    }
    if ((uint64_t)sg.WY * sg.WZ * 8 >= (1u << 24) || (int64_t)TX * TY * TZ < 256) return false;
    nthreads = TY * TZ;
    sg.nx = g.nx; sg.ny = g.ny; sg.nz = g.nz;
    sg.TX = TX; sg.TY = TY; sg.TZ = TZ;
    sg.MX = EX; sg.MY = EY; sg.MZ = EZ;
    sg.win_cells = (uint32_t)sg.WX * sg.WY * sg.WZ;
    sg.iw_chunks = sg.iw_bytes = 0;
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

static int device_cus() {
    static std::atomic<int> cus{0};
    int c = cus.load();
    if (c <= 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0)
            c = 256;
        cus = c;
    }
    return c;
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
    // the row-mapped kernel: one channel, or several without d_u (several with d_u: the geometry-once form below)
    int row_nt = 0;
    if (g_shear_row && shear_nt >= 1024 && (nc == 1 || !need_u) && make_shear_row(sg, g, nn, smem, row_nt)) {
#define LAGO_SHEAR_ROW(NU, UN, B)                                                                                 \
    do {                                                                                                          \
        auto k = splat_row_kernel<NU, UN, B>;                                                                     \
        if (smem > 64 * 1024) {                                                                                   \
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)smem);                                                                   \
            if (e != hipSuccess) return fail_hip(e, "interp_backward (row-mapped splat)");                        \
        }                                                                                                         \
        hipLaunchKernelGGL(k, dim3(sg.total), dim3(row_nt), smem, s, d_I, d_u, go, I, u, dt, nc, sg, umode, addgo); \
    } while (0)
        if (need_u) {
            if (unit_step) { if (bc) LAGO_SHEAR_ROW(true, true, true); else LAGO_SHEAR_ROW(true, true, false); }
            else { if (bc) LAGO_SHEAR_ROW(true, false, true); else LAGO_SHEAR_ROW(true, false, false); }
        } else {
            if (unit_step) { if (bc) LAGO_SHEAR_ROW(false, true, true); else LAGO_SHEAR_ROW(false, true, false); }
            else { if (bc) LAGO_SHEAR_ROW(false, false, true); else LAGO_SHEAR_ROW(false, false, false); }
        }
#undef LAGO_SHEAR_ROW
        note_path(LP_SPLAT_SHEAR_ROW);
        return finish_launch(s, "interp_backward (row-mapped splat)");
    }
    // d_u wanted, whole z rows of at most 160 voxels: the persistent, software-pipelined image-window kernel
    // (several channels / running d_u sums keep more per voxel: three voxels per thread instead of four)
    if (need_u && g_shear_pp && shear_nt >= 1024 && !((uintptr_t)I & 15u) &&
        make_shear_pp(sg, g, nn, smem, (nc > 1 || umode != 0) ? 3072 : 4096)) {
        const bool mcf = nc > 1 || umode != 0;
        const int vpl = sg.tile_vox <= 2048u ? 2 : (sg.tile_vox <= 3072u ? 3 : 4);
        const int parts = (sg.WZ + 63) / 64;
        const uint32_t cus = (uint32_t)device_cus();
        const uint32_t grid = (sg.total < cus ? sg.total : cus) & ~7u;
#define LAGO_SHEAR_PP_K(UN, B, V, M) (parts == 1 ? splat_shear_pp_kernel<UN, B, V, M, 1> : parts == 2 ? splat_shear_pp_kernel<UN, B, V, M, 2> : splat_shear_pp_kernel<UN, B, V, M, 3>)
#define LAGO_SHEAR_PP(UN, B)                                                                                      \
    do {                                                                                                          \
        auto k = mcf ? (vpl == 2 ? LAGO_SHEAR_PP_K(UN, B, 2, true) : vpl == 3 ? LAGO_SHEAR_PP_K(UN, B, 3, true) : LAGO_SHEAR_PP_K(UN, B, 4, true)) \
                     : (vpl == 2 ? LAGO_SHEAR_PP_K(UN, B, 2, false) : vpl == 3 ? LAGO_SHEAR_PP_K(UN, B, 3, false) : LAGO_SHEAR_PP_K(UN, B, 4, false)); \
        if (smem > 64 * 1024) {                                                                                   \
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)smem);                                                                   \
            if (e != hipSuccess) return fail_hip(e, "interp_backward (pipelined image-window splat)");            \
        }                                                                                                         \
        hipLaunchKernelGGL(k, dim3(grid), dim3(1024), smem, s, d_I, d_u, go, I, u, dt, nc, sg, umode, addgo);     \
    } while (0)
        if (unit_step) { if (bc) LAGO_SHEAR_PP(true, true); else LAGO_SHEAR_PP(true, false); }
        else { if (bc) LAGO_SHEAR_PP(false, true); else LAGO_SHEAR_PP(false, false); }
#undef LAGO_SHEAR_PP
#undef LAGO_SHEAR_PP_K
        note_path(LP_SPLAT_SHEAR_PP);
        return finish_launch(s, "interp_backward (pipelined image-window splat)");
    }
    // d_u wanted (any channel count): corners through an LDS window of I, every operand of a tile requested up front
    if (need_u && g_shear_iw && shear_nt >= 1024 && !((uintptr_t)I & 15u) && g.nx < 32768 && g.ny < 32768 &&
        make_shear(sg, g, nn, smem, 2048, true) && sg.iw_chunks <= 2048u && sg.nseg <= 64) {
        const bool one = sg.tile_vox <= 1024u;
#define LAGO_SHEAR_IW(UN, B)                                                                                      \
    do {                                                                                                          \
        auto k = one ? splat_shear_iw_kernel<1024, UN, B, 1, 8> : splat_shear_iw_kernel<1024, UN, B, 2, 8>;       \
        if (smem > 64 * 1024) {                                                                                   \
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)smem);                                                                   \
            if (e != hipSuccess) return fail_hip(e, "interp_backward (image-window splat)");                      \
        }                                                                                                         \
        hipLaunchKernelGGL(k, dim3(sg.total), dim3(1024), smem, s, d_I, d_u, go, I, u, dt, nc, sg, umode, addgo);  \
    } while (0)
        if (unit_step) { if (bc) LAGO_SHEAR_IW(true, true); else LAGO_SHEAR_IW(true, false); }
        else { if (bc) LAGO_SHEAR_IW(false, true); else LAGO_SHEAR_IW(false, false); }
#undef LAGO_SHEAR_IW
        note_path(LP_SPLAT_SHEAR_IW);
        return finish_launch(s, "interp_backward (image-window splat)");
    }
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
                     hipStream_t s) {
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
                                     bool, hipStream_t);
template int affine_splat_lds<double>(double *, const double *, const double *, const double *, int, int64_t,
                                      const Geom &, bool, hipStream_t);
template int regrid_splat_lds<float>(float *, const float *, int64_t, const Geom &, const Geom &, const double *,
                                     const double *, hipStream_t);
template int regrid_splat_lds<double>(double *, const double *, int64_t, const Geom &, const Geom &, const double *,
                                      const double *, hipStream_t);

}  // namespace lago

#ifdef LAGO_PROFILING
extern "C" void lago_debug_splat_stamps(void *buf) { (void)hipMemcpyToSymbol(HIP_SYMBOL(lago::g_dev_stamps), &buf, sizeof(void *)); }
extern "C" void lago_debug_splat_skip(int mask) { (void)hipMemcpyToSymbol(HIP_SYMBOL(lago::g_dev_splat_skip), &mask, sizeof(int)); }
#endif
extern "C" {
// Tuning hook (bench / tests): tile TX, TY, TZ (0 = auto), window margins, threads per workgroup.
// Affects speed only, never results.
void lago_set_splat_mc(int on) { lago::g_splat_mc = on; }
// sheared-window float32 splat: on/off, tile TX TY TZ (0 = auto), margins, threads per workgroup.  Speed only.
void lago_set_splat_shear_mc(int mode) {
    lago::g_shear_row = mode >= 5 ? 1 : 0;         // 5: the row-mapped kernel where it applies
    lago::g_shear_pp = mode >= 4 ? 1 : 0;          // 4: the persistent pipelined image-window kernel where it applies
    lago::g_shear_iw = mode >= 3 ? 1 : 0;          // 3: the one-shot image-window kernel where it applies
    lago::g_shear_mc = mode >= 3 ? 2 : mode;
}
void lago_set_splat_shear(int on, int tx, int ty, int tz, int mx, int my, int mz, int nthreads) {
    lago::g_shear_on = on;
    lago::g_shear_cfg.set({tx, ty, tz, mx, my, mz});
    lago::g_shear_nt = nthreads;
}
void lago_set_splat_tile(int tx, int ty, int tz, int ex, int ey, int ez, int nthreads) {
    lago::g_tile_cfg.set({tx, ty, tz, ex, ey, ez, nthreads});
}
}
