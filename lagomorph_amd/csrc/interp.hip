// Free-form interpolation and its adjoint (splat) -- gfx950 HIP kernels.
//
// Replaces cuda/interp.cu of the reference: interp_kernel_{2,3}d (:16-78),
// interp_kernel_backward_{2,3}d (:132-244) and
// interp_hessian_diagonal_image_kernel_2d (:317-349).  One lane per output
// voxel, last axis fastest: u / grad_out / out / d_u stream as 256-byte
// wavefront rows, the 8 (4) lerp corners are gathers served by L1/L2.
#include "common.hpp"
#include "gather_window.hpp"

#ifndef LAGO_NT_INTERPW_LD
#define LAGO_NT_INTERPW_LD 1   // u of the window kernel (C >= 2) is read once: interp_forward C = 3 148 -> 131 us
#endif
#ifndef LAGO_NT_INTERP_LD
#define LAGO_NT_INTERP_LD 0
#endif
#ifndef LAGO_NT_INTERP_ST
#define LAGO_NT_INTERP_ST 1   // forward outputs non-temporal: C = 1 forward 82 -> 73 us at 8 x 128^3 (profiles/r04_cache_policy.md)
#endif

namespace lago {

// ------------------------------------------------------------------ forward

template <typename R, int DIM, bool BC>
__global__ __launch_bounds__(kBlock) void interp_fwd_kernel(R *__restrict__ out, const R *__restrict__ I,
                                                            const R *__restrict__ u, double dt, int nc, Geom g) {
    const Vox v = locate(g);
    if (!v.valid) return;
    const size_t nv = g.nvox;
    const R *un = u + (size_t)v.n * DIM * nv + v.s;
    const R *In = BC ? I : I + (size_t)v.n * nc * nv;
    R *on = out + (size_t)v.n * nc * nv + v.s;
    if (DIM == 3) {
        R hx = sample_pos<R>(v.i, dt, un[0]);
        R hy = sample_pos<R>(v.j, dt, un[nv]);
        R hz = sample_pos<R>(v.k, dt, un[2 * nv]);
        Lerp3<R> L;
        L.setup(hx, hy, hz, g.nx, g.ny, g.nz);
        for (int c = 0; c < nc; ++c) on[(size_t)c * nv] = L.value(In + (size_t)c * nv);
    } else {
        R hx = sample_pos<R>(v.j, dt, un[0]);
        R hy = sample_pos<R>(v.k, dt, un[nv]);
        Lerp2<R> L;
        L.setup(hx, hy, g.ny, g.nz);
        for (int c = 0; c < nc; ++c) on[(size_t)c * nv] = L.value(In + (size_t)c * nv);
    }
}

// Unrolled 3D forward: a workgroup processes U slabs of 256 consecutive voxels, each lane one
// voxel of every slab.  The scalar kernel above is latency-bound (two dependent memory round trips
// per wave with 256 bytes in flight); giving each lane U *consecutive* voxels instead (16-byte
// vectors) fixes that but spreads a wave's gather over 1 KiB, and PMC counters showed the kernel
// then bound by L1 (TCP) line accesses: 652 per wave, one per clock per CU = the kernel time.
// With slab-interleaved voxels every load and every gather of a wave covers ~256 contiguous bytes
// (a few L1 lines), and the U * 4 pair gathers of a channel are still issued back to back.
template <typename R, bool BC, int U, bool UNIT>
__global__ __launch_bounds__(kBlock) void interp_fwd3_unroll_kernel(R *__restrict__ out, const R *__restrict__ I,
                                                                    const R *__restrict__ u, double dt, int nc,
                                                                    Geom g, uint32_t nbx_u, uint32_t nblocks_u) {
    const uint32_t Lb = block_order(blockIdx.x, nblocks_u, g.rev);
    const uint32_t n = Lb / nbx_u;  // uniform: scalar division
    const uint32_t bx = Lb - n * nbx_u;
    const size_t nv = g.nvox;
    const R *un = u + (size_t)n * 3 * nv;
    const R *In = BC ? I : I + (size_t)n * nc * nv;
    R *on = out + (size_t)n * nc * nv;
    uint32_t s[U];
    bool ok[U];
    R ux[U], uy[U], uz[U];
#pragma unroll
    for (int e = 0; e < U; ++e) {
        s[e] = (bx * U + e) * kBlock + threadIdx.x;
        ok[e] = s[e] < g.nvox;
        if (!ok[e]) s[e] = 0;
        ux[e] = ld_pol<LAGO_NT_INTERP_LD>(un + s[e]);
        uy[e] = ld_pol<LAGO_NT_INTERP_LD>(un + nv + s[e]);
        uz[e] = ld_pol<LAGO_NT_INTERP_LD>(un + 2 * nv + s[e]);
    }
    Lerp3<R, false> L[U];  // nz >= 2 guaranteed by the host: no per-sample thin branch
    uint32_t ci = 0, cj = 0, ck = 0;
    const uint32_t qj = (uint32_t)kBlock / (uint32_t)g.nz, rk = (uint32_t)kBlock % (uint32_t)g.nz;  // uniform
#pragma unroll
    for (int e = 0; e < U; ++e) {
        // (i, j, k) of slab e: one fast division for e = 0, then +256 voxels per slab as
        // (+qj rows, +rk voxels) with at most one carry each (host guarantees qj + 1 < ny)
        if (e == 0) {
            ci = g.dyz.div(s[0]);
            const uint32_t r = s[0] - ci * (uint32_t)(g.ny * g.nz);
            cj = g.dz.div(r);
            ck = r - cj * (uint32_t)g.nz;
        } else {
            ck += rk;
            cj += qj;
            if (ck >= (uint32_t)g.nz) { ck -= g.nz; ++cj; }
            if (cj >= (uint32_t)g.ny) { cj -= g.ny; ++ci; }
        }
        const uint32_t i = ci, j = cj, k = ck;
        L[e].setup(sample_pos_t<R, UNIT>((int)i, dt, ux[e]), sample_pos_t<R, UNIT>((int)j, dt, uy[e]),
                   sample_pos_t<R, UNIT>((int)k, dt, uz[e]), g.nx, g.ny, g.nz);
    }
    for (int c = 0; c < nc; ++c) {
        const R *Ic = In + (size_t)c * nv;
        R o[U];
#pragma unroll
        for (int e = 0; e < U; ++e) o[e] = L[e].value(Ic);
#pragma unroll
        for (int e = 0; e < U; ++e)
            if (ok[e]) st_pol<LAGO_NT_INTERP_ST>(&on[(size_t)c * nv + s[e]], o[e]);
    }
}

// LDS-window variant (gather_window.hpp; compose3_window_kernel of fused.hip is the three-channel form with the
// axpy): the nc channels of I go through one 48 KB window in turn, the displacement stays in registers.  Same
// expressions as Lerp3: bit-identical to interp_fwd3_unroll_kernel; samples whose corners leave the window take them
// with that kernel's pair gathers, lane by lane.
template <int NT, int U, bool UNIT, bool BC>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void interp3_window_kernel(
    float *__restrict__ out, const float *__restrict__ I, const float *__restrict__ u, double dt, int nc, Geom g, GWGrid w) {
    extern __shared__ float gwin[];
    constexpr int XS = NT / (32 * GW::TY);
    static_assert(XS * U == GW::TX && NT % (32 * GW::TY) == 0, "tile shape");
    const GWTile tl = gw_tile(w, g.rev);
    const size_t nv = g.nvox;
    const uint32_t plane = g.nvox * 4u;
    const float *un = u + (size_t)tl.n * 3 * nv;
    const float *In = BC ? I : I + (size_t)tl.n * nc * nv;
    float *on = out + (size_t)tl.n * nc * nv;
    const int lz = threadIdx.x & 31, ly = (threadIdx.x >> 5) & (GW::TY - 1), lxb = threadIdx.x / (32 * GW::TY);
    const int j = tl.y0 + ly, k = tl.z0 + lz;

    GWOrigin o;
    {
        const int cxi = min(tl.x0 + GW::TX / 2, g.nx - 1), cyi = min(tl.y0 + GW::TY / 2, g.ny - 1),
                  czi = min(tl.z0 + GW::TZ / 2, g.nz - 1);
        const uint32_t cs = ((uint32_t)cxi * (uint32_t)g.ny + (uint32_t)cyi) * (uint32_t)g.nz + (uint32_t)czi;
        o = gw_origin(tl, g, lg_floor(sample_pos_t<float, UNIT>(cxi, dt, un[cs])),
                      lg_floor(sample_pos_t<float, UNIT>(cyi, dt, un[nv + cs])),
                      lg_floor(sample_pos_t<float, UNIT>(czi, dt, un[2 * nv + cs])));
    }
    GWLoader<NT> ld;
    ld.plan(o, g);
    ld.issue(In, plane, gwin);

    const bool row_ok = j < g.ny && k < g.nz;
    const uint32_t off0 = (((uint32_t)(tl.x0 + lxb) * (uint32_t)g.ny + (uint32_t)j) * (uint32_t)g.nz + (uint32_t)k) * 4u;
    const uint32_t estep = (uint32_t)XS * (uint32_t)g.ny * (uint32_t)g.nz * 4u;
    auto voff = [&](int e) { return row_ok && tl.x0 + lxb + XS * e < g.nx ? off0 + (uint32_t)e * estep : GW::kOutside; };
    GWLerp L[U];
    uint32_t outm = 0;  // bit e: sample e has a corner outside the window
    float uu[3][U];
    {
#pragma unroll
        for (int e = 0; e < U; ++e) {
#pragma unroll
            for (int d = 0; d < 3; ++d)
                uu[d][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(make_rsrc(un + (size_t)d * nv, plane), voff(e), 0, LAGO_NT_INTERPW_LD ? 2 : 0));
        }
#pragma unroll
        for (int e = 0; e < U; ++e) {
            const int i = tl.x0 + lxb + XS * e;
            const bool in = L[e].setup(sample_pos_t<float, UNIT>(i, dt, uu[0][e]), sample_pos_t<float, UNIT>(j, dt, uu[1][e]),
                                       sample_pos_t<float, UNIT>(k, dt, uu[2][e]), g, o);
            outm |= (in | (voff(e) == GW::kOutside)) ? 0u : 1u << e;
            __builtin_amdgcn_sched_barrier(0);  // one sample at a time (registers), as in compose3_window_kernel
        }
    }
    auto channels = [&](auto stray) {
        constexpr bool STRAY = decltype(stray)::value;
        for (int c = 0; c < nc; ++c) {
            if (c > 0) {
                // channel c's window must have landed: in this rolled loop the LDS-direct loads were issued on the other
                // side of the back edge, so the wait is spelled out (in compose3_window_kernel's unrolled loop hipcc
                // inserts it in front of the barrier by itself)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            const float *Ic = In + (size_t)c * nv;
            const BufRsrc ro = make_rsrc(on + (size_t)c * nv, plane);
#pragma unroll
            for (int e = 0; e < U; ++e) {
                float val = L[e].value(gwin);
                if constexpr (STRAY) {
                    const bool mine = outm & (1u << e);
                    if (__builtin_amdgcn_ballot_w64(mine) != 0) {  // wave-uniform
                        if (mine) {
                            const int i = tl.x0 + lxb + XS * e;
                            Lerp3<float, false> P;
                            P.setup(sample_pos_t<float, UNIT>(i, dt, uu[0][e]), sample_pos_t<float, UNIT>(j, dt, uu[1][e]),
                                    sample_pos_t<float, UNIT>(k, dt, uu[2][e]), g.nx, g.ny, g.nz);
                            val = P.value(Ic);
                        }
                    }
                }
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, val), ro, voff(e), 0, LAGO_NT_INTERP_ST ? 2 : 0);
            }
            if (c + 1 < nc) {
                __syncthreads();  // everyone has read channel c: the window is free
                ld.issue(Ic + nv, plane, gwin);
            }
        }
    };
    // channel 0's window has landed behind this barrier (each wave waits for its own LDS-direct loads first)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (__syncthreads_or(outm != 0))
        channels(std::true_type{});
    else
        channels(std::false_type{});
}

// (explicit instantiations: hipcc 7.2 drops the host stub of a variant that is only named in the `else` of a launch)
template __global__ void interp3_window_kernel<GW::NT, 8, true, true>(float *, const float *, const float *, double, int, Geom, GWGrid);
template __global__ void interp3_window_kernel<GW::NT, 8, true, false>(float *, const float *, const float *, double, int, Geom, GWGrid);
template __global__ void interp3_window_kernel<GW::NT, 8, false, true>(float *, const float *, const float *, double, int, Geom, GWGrid);
template __global__ void interp3_window_kernel<GW::NT, 8, false, false>(float *, const float *, const float *, double, int, Geom, GWGrid);

extern std::atomic<int> g_gather_window;   // fused.hip

template <typename R>
static bool interp_window_launch(R *out, const R *I, const R *u, double dt, int nc, bool bc, const Geom &g, int64_t nn,
                                 hipStream_t s) {
    if constexpr (sizeof(R) == 4) {
        GWGrid w;
        if (!make_gwgrid(w, g, nn) || ((uintptr_t)I & 15u)) return false;
        constexpr int NT = GW::NT, U = 8;
        const size_t smem = GW::lds_bytes<NT>();
        const bool unit = unit_dt<R>(dt);
#define LAGO_IW(UN, B) \
    hipLaunchKernelGGL((interp3_window_kernel<NT, U, UN, B>), dim3(w.total), dim3(NT), smem, s, out, I, u, dt, nc, g, w)
        if (unit) { if (bc) LAGO_IW(true, true); else LAGO_IW(true, false); }
        else { if (bc) LAGO_IW(false, true); else LAGO_IW(false, false); }
#undef LAGO_IW
        note_path(LP_GATHER_WINDOW);
        return true;
    }
    return false;
}

// ------------------------------------------------------------------ backward, global atomics

template <typename R, int DIM, bool BC, bool NEED_I, bool NEED_U>
__global__ __launch_bounds__(kBlock) void interp_bwd_kernel(R *__restrict__ d_I, R *__restrict__ d_u,
                                                            const R *__restrict__ go, const R *__restrict__ I,
                                                            const R *__restrict__ u, double dt, int nc, Geom g,
                                                            int umode, R addgo) {
    const Vox v = locate(g);
    if (!v.valid) return;
    const size_t nv = g.nvox;
    const R *un = u + (size_t)v.n * DIM * nv + v.s;
    const R *In = BC ? I : I + (size_t)v.n * nc * nv;
    R *dIn = BC ? d_I : d_I + (size_t)v.n * nc * nv;
    const R *gon = go + (size_t)v.n * nc * nv + v.s;
    // start value of the thread-owned d_u sum: zero (the reference), the caller's d_u, or addgo * grad_out[component]
    R u0[3] = {(R)0, (R)0, (R)0};
    if (NEED_U && umode) {
        const R *dup = d_u + (size_t)v.n * DIM * nv + v.s;
#pragma unroll
        for (int d = 0; d < DIM; ++d) u0[d] = umode == 1 ? dup[(size_t)d * nv] : addgo * gon[(size_t)d * nv];
    }
    if (DIM == 3) {
        R hx = sample_pos<R>(v.i, dt, un[0]);
        R hy = sample_pos<R>(v.j, dt, un[nv]);
        R hz = sample_pos<R>(v.k, dt, un[2 * nv]);
        Splat3<R> S;
        Lerp3<R> L;
        if (NEED_I) S.setup(hx, hy, hz, g.nx, g.ny, g.nz);
        if (NEED_U) L.setup(hx, hy, hz, g.nx, g.ny, g.nz);
        R ax = u0[0], ay = u0[1], az = u0[2];
        for (int c = 0; c < nc; ++c) {
            R diff = gon[(size_t)c * nv];
            if (NEED_I) {
                R *dIc = dIn + (size_t)c * nv;
#pragma unroll
                for (int q = 0; q < 8; ++q) atomic_add(dIc + S.o[q], S.w[q] * diff);
            }
            if (NEED_U) {
                R gx, gy, gz;
                L.grad(In + (size_t)c * nv, gx, gy, gz);
                diff = (R)((double)diff * dt);  // cuda/interp.cu:230
                ax = lg_fma(gx, diff, ax);
                ay = lg_fma(gy, diff, ay);
                az = lg_fma(gz, diff, az);
            }
        }
        if (NEED_U) {
            R *dun = d_u + (size_t)v.n * DIM * nv + v.s;
            dun[0] = ax;
            dun[nv] = ay;
            dun[2 * nv] = az;
        }
    } else {
        R hx = sample_pos<R>(v.j, dt, un[0]);
        R hy = sample_pos<R>(v.k, dt, un[nv]);
        Splat2<R> S;
        Lerp2<R> L;
        if (NEED_I) S.setup(hx, hy, g.ny, g.nz);
        if (NEED_U) L.setup(hx, hy, g.ny, g.nz);
        R ax = u0[0], ay = u0[1];
        for (int c = 0; c < nc; ++c) {
            R diff = gon[(size_t)c * nv];
            if (NEED_I) {
                R *dIc = dIn + (size_t)c * nv;
#pragma unroll
                for (int q = 0; q < 4; ++q) atomic_add(dIc + S.o[q], S.w[q] * diff);
            }
            if (NEED_U) {
                R gx, gy;
                L.grad(In + (size_t)c * nv, gx, gy);
                diff = (R)((double)diff * dt);  // cuda/interp.cu:171
                ax = lg_fma(gx, diff, ax);
                ay = lg_fma(gy, diff, ay);
            }
        }
        if (NEED_U) {
            R *dun = d_u + (size_t)v.n * DIM * nv + v.s;
            dun[0] = ax;
            dun[nv] = ay;
        }
    }
}

// ------------------------------------------------------------------ backward, 2D, LDS-privatised
//
// interp_kernel_backward_2d (cuda/interp.cu:132-183 + atomicSplat include/interp.h:404-424): the reference's whole
// test-suite and the 2D lagomorph.affine path splat with four global float atomics per pixel-channel, which the memory
// side retires at 1.3 TB/s of added bytes.  Same scheme as the 3D kernels of splat.hip: a workgroup owns a TH x TW tile
// of source pixels (lanes along the row), accumulates their four corner contributions in a float64 LDS window placed
// at tile origin + the displacement probed at the tile centre - margin (its first column a multiple of 16: 64-byte
// aligned flush rows), and flushes the window with one global atomic per touched cell.  Corners that miss the window
// take the reference's clamped global atomics, so any displacement is handled.  d_u (biLerp_grad, include/interp.h:
// 128-203) is produced by the same pass, accumulated per pixel in registers over the channels in ascending order:
// bit-identical to interp_bwd_kernel.
struct Splat2Geom {
    int H, W;                 // image (= Geom::ny, nz)
    int TH, TW, WH, WW, MH, MW;
    uint32_t nth, ntw, tiles_per_item, total;
    int rev;
    FastDiv d_tiles, d_tw;
};
constexpr int kS2T = 256, kS2V = 4;   // threads per workgroup, pixels per thread (TH * TW = kS2T * kS2V, TW = 64)

template <typename R, bool BC, bool NEED_U>
__global__ __launch_bounds__(kS2T) void splat2d_lds_kernel(R *__restrict__ d_I, R *__restrict__ d_u, const R *__restrict__ go,
                                                           const R *__restrict__ I, const R *__restrict__ u, double dt, int nc,
                                                           Splat2Geom sg, int umode, R addgo) {
    extern __shared__ __align__(16) unsigned char lago_s2[];
    double *win = reinterpret_cast<double *>(lago_s2);
    const int H = sg.H, W = sg.W;
    const size_t nv = (size_t)H * W;
    const uint32_t L = block_order(blockIdx.x, sg.total, sg.rev);
    const uint32_t n = sg.d_tiles.div(L);
    const uint32_t r = L - n * sg.tiles_per_item;
    const uint32_t by = sg.d_tw.div(r), bx = r - by * sg.ntw;
    const int y0 = (int)by * sg.TH, x0 = (int)bx * sg.TW;
    const R *un = u + (size_t)n * 2 * nv;
    const R *In = BC ? I : I + (size_t)n * nc * nv;
    R *dIn = BC ? d_I : d_I + (size_t)n * nc * nv;
    const R *gon = go + (size_t)n * nc * nv;
    R *dun = NEED_U ? d_u + (size_t)n * 2 * nv : nullptr;
    // window placement (speed only): the displacement of the tile's centre pixel
    const int weh = min(sg.WH, H), wew = min(sg.WW, W);
    int wy0, wx0;
    {
        const int cy = min(y0 + sg.TH / 2, H - 1), cx = min(x0 + sg.TW / 2, W - 1);
        const float fdt = (float)dt;
        const int oy = y0 + (int)floorf(fdt * (float)un[(size_t)cy * W + cx]) - sg.MH;
        const int ox = x0 + (int)floorf(fdt * (float)un[nv + (size_t)cy * W + cx]) - sg.MW;
        wy0 = max(0, min(oy, H - weh));
        wx0 = max(0, min(ox & ~15, W - wew));
    }
    const int WWp = sg.WW;   // window row pitch (cells)
    constexpr uint32_t NOWIN = 0xffffffffu;
    // this thread's pixels: q = t + e * kS2T -> (row q / 64, column q % 64): a wave is one row piece (TW = 64)
    bool live[kS2V];
    size_t sv[kS2V];
    Splat2<R> S[kS2V];
    uint32_t wl[kS2V][4];   // window cell of every (clamped) corner, or NOWIN: the same for every channel
    Lerp2<R> Lq[kS2V];
    R ax[kS2V], ay[kS2V];
#pragma unroll
    for (int e = 0; e < kS2V; ++e) {
        const int q = (int)threadIdx.x + e * kS2T;
        const int pj = y0 + (q >> 6), pk = x0 + (q & 63);
        live[e] = pj < H && pk < W;
        sv[e] = live[e] ? (size_t)pj * W + pk : 0;
        const R hx = sample_pos<R>(pj, dt, un[sv[e]]);
        const R hy = sample_pos<R>(pk, dt, un[nv + sv[e]]);
        S[e].setup(hx, hy, H, W);
        {   // the corners' rows / columns again (Splat2 keeps the linear index only): same clamps, corner order a outer, b inner
            const int fx = lg_floor(hx), fy = lg_floor(hy);
            const uint32_t ly[2] = {(uint32_t)(clamp1(fx, H) - wy0), (uint32_t)(clamp1(fx + 1, H) - wy0)};
            const uint32_t lx[2] = {(uint32_t)(clamp1(fy, W) - wx0), (uint32_t)(clamp1(fy + 1, W) - wx0)};
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const uint32_t yy = ly[q4 >> 1], xx = lx[q4 & 1];
                wl[e][q4] = (yy < (uint32_t)weh && xx < (uint32_t)wew) ? yy * (uint32_t)WWp + xx : NOWIN;
            }
        }
        if (NEED_U) Lq[e].setup(hx, hy, H, W);
        ax[e] = ay[e] = (R)0;
        if (NEED_U && umode) {   // start of the thread-owned d_u sum: the caller's d_u, or addgo * grad_out[component]
            ax[e] = umode == 1 ? dun[sv[e]] : addgo * gon[sv[e]];
            ay[e] = umode == 1 ? dun[nv + sv[e]] : addgo * gon[nv + sv[e]];
        }
    }
    for (int c = 0; c < nc; ++c) {
        for (int f = threadIdx.x; f < sg.WH * WWp; f += kS2T) win[f] = 0.0;
        __syncthreads();
        R *dIc = dIn + (size_t)c * nv;
        const R *gc = gon + (size_t)c * nv;
#pragma unroll
        for (int e = 0; e < kS2V; ++e) {
            if (!live[e]) continue;
            R diff = gc[sv[e]];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const R w = S[e].w[q] * diff;                      // include/interp.h:404-424
                if (wl[e][q] != NOWIN)
                    __hip_atomic_fetch_add(win + wl[e][q], (double)w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else
                    atomic_add(dIc + S[e].o[q], w);
            }
            if (NEED_U) {
                R gx, gy;
                Lq[e].grad(In + (size_t)c * nv, gx, gy);
                diff = (R)((double)diff * dt);  // cuda/interp.cu:171
                ax[e] = lg_fma(gx, diff, ax[e]);
                ay[e] = lg_fma(gy, diff, ay[e]);
            }
        }
        __syncthreads();
        // flush: one wave per window row, lanes along the row
        {
            const int lane = threadIdx.x & 63;
            const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
            for (int row = wave; row < weh; row += kS2T / 64) {
                R *grow = dIc + (size_t)(wy0 + row) * W + wx0;
                for (int col = lane; col < wew; col += 64) {
                    const double acc = win[row * WWp + col];
                    if (acc != 0.0) atomic_add(grow + col, (R)acc);
                }
            }
        }
        __syncthreads();
    }
    if (NEED_U) {
#pragma unroll
        for (int e = 0; e < kS2V; ++e)
            if (live[e]) {
                dun[sv[e]] = ax[e];
                dun[nv + sv[e]] = ay[e];
            }
    }
}

// returns false when the shape is left to the global-atomics kernel
template <typename R>
static bool interp_backward_2d_lds(R *d_I, R *d_u, const R *go, const R *I, const R *u, double dt, int nc, int64_t nn,
                                   const Geom &g, bool bc, bool need_u, int umode, double addgo, hipStream_t s) {
    if (g.ny < 16 || g.nz < 64 || (int64_t)g.ny * g.nz < 8192) return false;   // not worth a window
    Splat2Geom sg;
    sg.H = g.ny; sg.W = g.nz;
    sg.TW = 64; sg.TH = kS2T * kS2V / 64;   // (the kernel's pixel mapping assumes 64-pixel tile rows)
    sg.MH = 2; sg.MW = 4;
    sg.WH = sg.TH + 1 + 2 * sg.MH;
    sg.WW = ((sg.TW + 1 + 2 * sg.MW + 15 + 15) / 16) * 16;   // + 15: the first column is aligned down
    sg.nth = (uint32_t)((g.ny + sg.TH - 1) / sg.TH);
    sg.ntw = (uint32_t)((g.nz + sg.TW - 1) / sg.TW);
    sg.tiles_per_item = sg.nth * sg.ntw;
    const int64_t total = (int64_t)sg.tiles_per_item * nn;
    if (total <= 0 || total >= (1ll << 31)) return false;
    sg.total = (uint32_t)total;
    sg.rev = g.rev;
    sg.d_tiles = FastDiv(sg.tiles_per_item);
    sg.d_tw = FastDiv(sg.ntw);
    const size_t smem = (size_t)sg.WH * sg.WW * sizeof(double);
#define LAGO_S2(B, NU) \
    hipLaunchKernelGGL((splat2d_lds_kernel<R, B, NU>), dim3(sg.total), dim3(kS2T), smem, s, d_I, d_u, go, I, u, dt, nc, sg, umode, (R)addgo)
    if (bc) { if (need_u) LAGO_S2(true, true); else LAGO_S2(true, false); }
    else { if (need_u) LAGO_S2(false, true); else LAGO_S2(false, false); }
#undef LAGO_S2
    note_path(LP_SPLAT_2D);
    return true;
}

// ------------------------------------------------------------------ Hessian diagonal (2D)

template <typename R>
__global__ __launch_bounds__(kBlock) void interp_hessdiag_kernel(R *__restrict__ out, const R *__restrict__ u,
                                                                 double dt, int nc, Geom g) {
    const Vox v = locate(g);
    if (!v.valid) return;
    const size_t nv = g.nvox;
    const R *un = u + (size_t)v.n * 2 * nv + v.s;
    R x = sample_pos<R>(v.j, dt, un[0]);
    R y = sample_pos<R>(v.k, dt, un[nv]);
    Lerp2<R> L;
    L.setup(x, y, g.ny, g.nz);
    R omt = (R)1.f - L.t, omu = (R)1.f - L.u;
    R w0 = omt * omu, w1 = L.t * omu, w2 = L.t * L.u, w3 = omt * L.u;  // include/interp.h:522-525
    for (int c = 0; c < nc; ++c) {  // every (n, c) lands in plane 0 (cuda/interp.cu:342)
        atomic_add(out + L.o[0], w0 * w0);
        atomic_add(out + L.o[1], w1 * w1);
        atomic_add(out + L.o[2], w2 * w2);
        atomic_add(out + L.o[3], w3 * w3);
    }
}

// ------------------------------------------------------------------ host entry points

template <typename R>
static int interp_forward_impl(R *out, const R *I, const R *u, double dt, int dim, int64_t nn, int64_t nc,
                               int64_t nx, int64_t ny, int64_t nz, int bc, void *stream) {
    if (dim != 2 && dim != 3)
        return fail_invalid("Only two- and three-dimensional interpolation is supported");
    Geom g;
    if (nc < 0 || !make_geom(g, dim, nn, nx, ny, nz)) return fail_invalid("interp_forward: bad extent");
    if (g.nblocks == 0 || nc == 0) return LAGO_OK;  // empty batch / no channels: nothing to write
    if (!out || !I || !u) return fail_invalid("interp_forward: null pointer");
    hipStream_t s = (hipStream_t)stream;
    constexpr int U = 2;
    // several channels: through the LDS window (-11 ... -21 % at C = 3); ONE channel does not pay for the window's setup
    // and barrier (81 -> 79 us at 8 x 128^3, slower at 10-voxel deformations and for a broadcast image: tools/ab_interp_window.py)
    if (dim == 3 && g_interp_vec && g_gather_window && nc >= 2 && g.nvox >= 32768u &&
        interp_window_launch<R>(out, I, u, dt, (int)nc, bc != 0, g, nn, s))
        return finish_launch(s, "interp_forward");
    if (dim == 3 && g_interp_vec && g.nz >= 2 && kBlock / g.nz + 1 < g.ny && g.nvox >= 4u * U * kBlock) {
        const uint32_t nbx_u = (g.nvox + U * kBlock - 1) / (U * kBlock);
        const uint64_t nb = (uint64_t)nbx_u * (uint64_t)nn;
        if (nb < (1ull << 31)) {
#define LAUNCH_U(B, UN)                                                                                              \
    hipLaunchKernelGGL((interp_fwd3_unroll_kernel<R, B, U, UN>), dim3((uint32_t)nb), dim3(kBlock), 0, s, out, I, u, dt, \
                       (int)nc, g, nbx_u, (uint32_t)nb)
            const bool unit = unit_dt<R>(dt);
            if (bc) {
                if (unit) LAUNCH_U(true, true); else LAUNCH_U(true, false);
            } else {
                if (unit) LAUNCH_U(false, true); else LAUNCH_U(false, false);
            }
#undef LAUNCH_U
            note_path(LP_VECTOR_GATHER);
            return finish_launch(s, "interp_forward");
        }
    }
#define LAUNCH(D, B) \
    hipLaunchKernelGGL((interp_fwd_kernel<R, D, B>), dim3(g.nblocks), dim3(kBlock), 0, s, out, I, u, dt, (int)nc, g)
    if (dim == 3) {
        if (bc) LAUNCH(3, true); else LAUNCH(3, false);
    } else {
        if (bc) LAUNCH(2, true); else LAUNCH(2, false);
    }
#undef LAUNCH
    return finish_launch(s, "interp_forward");
}

template <typename R, int DIM, bool BC>
static void launch_bwd(R *d_I, R *d_u, const R *go, const R *I, const R *u, double dt, int nc, const Geom &g,
                       bool need_I, bool need_u, int umode, double addgo, hipStream_t s) {
#define LAUNCH(NI, NU)                                                                                          \
    hipLaunchKernelGGL((interp_bwd_kernel<R, DIM, BC, NI, NU>), dim3(g.nblocks), dim3(kBlock), 0, s, d_I, d_u, \
                       go, I, u, dt, nc, g, umode, (R)addgo)
    if (need_I && need_u) LAUNCH(true, true);
    else if (need_I) LAUNCH(true, false);
    else if (need_u) LAUNCH(false, true);
#undef LAUNCH
}

template <typename R>
int interp_backward_lds(R *d_I, R *d_u, const R *go, const R *I, const R *u, double dt, int nc, int64_t nn,
                        const Geom &g, bool bc, bool need_u, int umode, double addgo, hipStream_t s);  // splat.hip

template <typename R>
static int interp_backward_impl(R *d_I, R *d_u, const R *go, const R *I, const R *u, double dt, int dim,
                                int64_t nn, int64_t nc, int64_t nx, int64_t ny, int64_t nz, int bc, int need_I,
                                int need_u, void *stream, int umode = 0, double addgo = 0.0, int imode = 0) {
    if (dim != 2 && dim != 3)
        return fail_invalid("Only two- and three-dimensional interpolation is supported");
    if (umode < 0 || umode > 2 || (umode == 2 && nc != dim))
        return fail_invalid("interp_backward_fused: u_mode must be 0, 1 or 2 (2 needs as many channels as dimensions)");
    if (umode && !need_u) return fail_invalid("interp_backward_fused: u_mode != 0 needs need_u");
    Geom g;
    if (nc < 0 || !make_geom(g, dim, nn, nx, ny, nz, kBlock, true)) return fail_invalid("interp_backward: bad extent");
    hipStream_t s = (hipStream_t)stream;
    const size_t nI = (size_t)(bc ? 1 : nn) * nc * g.nvox;
    const size_t nu = (size_t)nn * dim * g.nvox;
    if ((nI && !d_I) || (nu && !d_u) || (nu && nc && (!go || !I || !u)))
        return fail_invalid("interp_backward: null pointer");
    // d_I is a scatter target (or unused): zero it -- unless the caller asked for the splat to be added onto its
    // contents (imode 1).  d_u is fully overwritten when needed.
    if (imode < 0 || imode > 1 || (imode == 1 && !need_I)) return fail_invalid("interp_backward_fused: bad i_mode");
    if (nI && imode == 0) LAGO_HIP_TRY(hipMemsetAsync(d_I, 0, nI * sizeof(R), s));
    if (!need_u && nu) LAGO_HIP_TRY(hipMemsetAsync(d_u, 0, nu * sizeof(R), s));
    if (g.nblocks == 0 || nc == 0 || !(need_I || need_u)) {
        if (need_u && nu && umode != 1) LAGO_HIP_TRY(hipMemsetAsync(d_u, 0, nu * sizeof(R), s));
        return finish_launch(s, "interp_backward");
    }
    if (dim == 3 && need_I && g_splat_mode >= 1) {
        int rc = interp_backward_lds<R>(d_I, d_u, go, I, u, dt, (int)nc, nn, g, bc != 0, need_u != 0, umode, addgo, s);
        if (rc != 1) return rc;  // 1 = shape not supported by the tiled kernel, fall through
    }
    if (dim == 2 && need_I && g_splat_mode >= 1 &&
        interp_backward_2d_lds<R>(d_I, d_u, go, I, u, dt, (int)nc, nn, g, bc != 0, need_u != 0, umode, addgo, s))
        return finish_launch(s, "interp_backward");
    note_path(LP_SPLAT_GLOBAL);
    if (dim == 3) {
        if (bc) launch_bwd<R, 3, true>(d_I, d_u, go, I, u, dt, (int)nc, g, need_I, need_u, umode, addgo, s);
        else launch_bwd<R, 3, false>(d_I, d_u, go, I, u, dt, (int)nc, g, need_I, need_u, umode, addgo, s);
    } else {
        if (bc) launch_bwd<R, 2, true>(d_I, d_u, go, I, u, dt, (int)nc, g, need_I, need_u, umode, addgo, s);
        else launch_bwd<R, 2, false>(d_I, d_u, go, I, u, dt, (int)nc, g, need_I, need_u, umode, addgo, s);
    }
    return finish_launch(s, "interp_backward");
}

template <typename R>
static int hessdiag_impl(R *out, const R *u, double dt, int64_t nI, int64_t nn, int64_t nc, int64_t nx, int64_t ny,
                         void *stream) {
    Geom g;
    if (nc < 0 || nI < 0 || !make_geom(g, 2, nn, nx, ny, 1, kBlock, true))
        return fail_invalid("interp_hessian_diagonal_image: bad extent");
    hipStream_t s = (hipStream_t)stream;
    size_t no = (size_t)nI * nc * g.nvox;
    if ((no && !out) || (no && g.nblocks && !u)) return fail_invalid("interp_hessian_diagonal_image: null pointer");
    if (no) LAGO_HIP_TRY(hipMemsetAsync(out, 0, no * sizeof(R), s));
    if (g.nblocks && nc && no)
        hipLaunchKernelGGL((interp_hessdiag_kernel<R>), dim3(g.nblocks), dim3(kBlock), 0, s, out, u, dt, (int)nc, g);
    return finish_launch(s, "interp_hessian_diagonal_image");
}

}  // namespace lago

extern "C" {
#define LAGO_DEFINE(REAL, SUF)                                                                                     \
    int lago_interp_forward##SUF(REAL *out, const REAL *I, const REAL *u, double dt, int dim, int64_t nn,         \
                                 int64_t nc, int64_t nx, int64_t ny, int64_t nz, int bc, void *stream) {          \
        return lago::interp_forward_impl<REAL>(out, I, u, dt, dim, nn, nc, nx, ny, nz, bc, stream);               \
    }                                                                                                              \
    int lago_interp_backward##SUF(REAL *d_I, REAL *d_u, const REAL *go, const REAL *I, const REAL *u, double dt,  \
                                  int dim, int64_t nn, int64_t nc, int64_t nx, int64_t ny, int64_t nz, int bc,    \
                                  int need_I, int need_u, void *stream) {                                         \
        return lago::interp_backward_impl<REAL>(d_I, d_u, go, I, u, dt, dim, nn, nc, nx, ny, nz, bc, need_I,      \
                                                need_u, stream);                                                  \
    }                                                                                                              \
    int lago_interp_backward_fused##SUF(REAL *d_I, REAL *d_u, const REAL *go, const REAL *I, const REAL *u,       \
                                        double dt, int dim, int64_t nn, int64_t nc, int64_t nx, int64_t ny,       \
                                        int64_t nz, int bc, int need_I, int i_mode, int u_mode, double addgo,     \
                                        void *stream) {                                                           \
        return lago::interp_backward_impl<REAL>(d_I, d_u, go, I, u, dt, dim, nn, nc, nx, ny, nz, bc, need_I, 1,   \
                                                stream, u_mode, addgo, i_mode);                                   \
    }                                                                                                              \
    int lago_interp_hessian_diagonal_image##SUF(REAL *out, const REAL *u, double dt, int64_t nI, int64_t nn,      \
                                                int64_t nc, int64_t nx, int64_t ny, void *stream) {               \
        return lago::hessdiag_impl<REAL>(out, u, dt, nI, nn, nc, nx, ny, stream);                                 \
    }
LAGO_DEFINE(float, _f32)
LAGO_DEFINE(double, _f64)
#undef LAGO_DEFINE
}
