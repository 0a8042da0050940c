// Fused geometry operators -- gfx950 HIP kernels.
//
// compose: out = ds*u + dt*interp(v, u, ds), i.e. deform.compose of the reference
// (/root/reference/lagomorph/deform.py:53-55), which there is one interp kernel
// plus three elementwise torch kernels (84 extra bytes per voxel of traffic for a
// 3-vector field).  The three roundings of the unfused expression are kept
// (fl(fl(ds*u) + fl(dt*I))), so the result is bit-identical to evaluating the
// reference formula with this library's interp.
#include <algorithm>
#include <type_traits>
#include <stdlib.h>

#include "common.hpp"
#include "stencil_tile.hpp"
#include "gather_window.hpp"

namespace lago {

template <typename R, int DIM>
__global__ __launch_bounds__(kBlock) void compose_kernel(R *__restrict__ out, const R *__restrict__ u,
                                                         const R *__restrict__ v, double ds, double dt, Geom g) {
    const Vox vx = locate(g);
    if (!vx.valid) return;
    const size_t nv = g.nvox;
    const R *un = u + (size_t)vx.n * DIM * nv + vx.s;
    const R *vn = v + (size_t)vx.n * DIM * nv;
    R *on = out + (size_t)vx.n * DIM * nv + vx.s;
    const R dsr = (R)ds, dtr = (R)dt;  // torch multiplies by the scalar rounded to the tensor dtype
    R uv[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) uv[d] = un[(size_t)d * nv];
    if (DIM == 3) {
        Lerp3<R> L;
        L.setup(sample_pos<R>(vx.i, ds, uv[0]), sample_pos<R>(vx.j, ds, uv[1]), sample_pos<R>(vx.k, ds, uv[2]),
                g.nx, g.ny, g.nz);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const R a = dsr * uv[c];
            const R b = dtr * L.value(vn + (size_t)c * nv);
            on[(size_t)c * nv] = a + b;
        }
    } else {
        Lerp2<R> L;
        L.setup(sample_pos<R>(vx.j, ds, uv[0]), sample_pos<R>(vx.k, ds, uv[1]), g.ny, g.nz);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const R a = dsr * uv[c];
            const R b = dtr * L.value(vn + (size_t)c * nv);
            on[(size_t)c * nv] = a + b;
        }
    }
}

// Unrolled 3D variant (see interp_fwd3_unroll_kernel in interp.hip): U slabs of 256 consecutive
// voxels per workgroup, one voxel of each slab per lane.
template <typename R, int U, bool UNIT>
__global__ __launch_bounds__(kBlock) void compose3_unroll_kernel(R *__restrict__ out, const R *__restrict__ u,
                                                                 const R *__restrict__ v, double ds, double dt, Geom g,
                                                                 uint32_t nbx_u, uint32_t nblocks_u) {
    const uint32_t Lb = block_order(blockIdx.x, nblocks_u, g.rev);
    const uint32_t n = Lb / nbx_u;
    const uint32_t bx = Lb - n * nbx_u;
    const size_t nv = g.nvox;
    const R *un = u + (size_t)n * 3 * nv;
    const R *vn = v + (size_t)n * 3 * nv;
    R *on = out + (size_t)n * 3 * nv;
    const R dsr = (R)ds, dtr = (R)dt;
    uint32_t s[U];
    bool ok[U];
    R uu[3][U];
#pragma unroll
    for (int e = 0; e < U; ++e) {
        s[e] = (bx * U + e) * kBlock + threadIdx.x;
        ok[e] = s[e] < g.nvox;
        if (!ok[e]) s[e] = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d) uu[d][e] = un[(size_t)d * nv + s[e]];
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
        L[e].setup(sample_pos_t<R, UNIT>((int)i, ds, uu[0][e]), sample_pos_t<R, UNIT>((int)j, ds, uu[1][e]),
                   sample_pos_t<R, UNIT>((int)k, ds, uu[2][e]), g.nx, g.ny, g.nz);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        R o[U];
#pragma unroll
        for (int e = 0; e < U; ++e) {
            const R a = dsr * uu[c][e];
            const R b = dtr * L[e].value(vn + (size_t)c * nv);
            o[e] = a + b;
        }
#pragma unroll
        for (int e = 0; e < U; ++e)
            if (ok[e]) on[(size_t)c * nv + s[e]] = o[e];
    }
}

// LDS-window variant (gather_window.hpp): the three channels of v go through one 48 KB window in turn.
// Same expressions as compose3_unroll_kernel; samples whose corners leave the window take them with that kernel's
// pair gathers.
#ifndef LAGO_NT_AD_LD
#define LAGO_NT_AD_LD 0
#endif
#ifndef LAGO_NT_AD_ST
#define LAGO_NT_AD_ST 1   // Ad_star output non-temporal: -3 % on the kernel (profiles/r04_cache_policy.md)
#endif
#ifndef LAGO_COMPOSE_AUX_LD
// the window compose reads u once and writes once: both non-temporal (2 = nt) -4.7 % on the kernel, -0.9 % on the
// shoot; either one alone: nothing (profiles/r04_cache_policy.md)
#define LAGO_COMPOSE_AUX_LD 2
#define LAGO_COMPOSE_AUX_ST 2
#endif
template <int NT, int U, bool UNIT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void compose3_window_kernel(
    float *__restrict__ out, const float *__restrict__ u, const float *__restrict__ v, double ds, double dt, Geom g, GWGrid w) {
    extern __shared__ float gwin[];
    constexpr int XS = NT / (32 * GW::TY);
    static_assert(XS * U == GW::TX && NT % (32 * GW::TY) == 0, "tile shape");  // 1024 x 4 has 64 VGPRs: measured slower
    const GWTile tl = gw_tile(w, g.rev);
    const size_t nv = g.nvox;
    const uint32_t plane = g.nvox * 4u;
    const float *un = u + (size_t)tl.n * 3 * nv;
    const float *vn = v + (size_t)tl.n * 3 * nv;
    float *on = out + (size_t)tl.n * 3 * nv;
    const float dsr = (float)ds, dtr = (float)dt;
    const int lz = threadIdx.x & 31, ly = (threadIdx.x >> 5) & (GW::TY - 1), lxb = threadIdx.x / (32 * GW::TY);
    const int j = tl.y0 + ly, k = tl.z0 + lz;

    // window origin from the tile's centre voxel (uniform loads), then start moving channel 0
    GWOrigin o;
    {
        const int cxi = min(tl.x0 + GW::TX / 2, g.nx - 1), cyi = min(tl.y0 + GW::TY / 2, g.ny - 1),
                  czi = min(tl.z0 + GW::TZ / 2, g.nz - 1);
        const uint32_t cs = ((uint32_t)cxi * (uint32_t)g.ny + (uint32_t)cyi) * (uint32_t)g.nz + (uint32_t)czi;
        o = gw_origin(tl, g, lg_floor(sample_pos_t<float, UNIT>(cxi, ds, un[cs])),
                      lg_floor(sample_pos_t<float, UNIT>(cyi, ds, un[nv + cs])),
                      lg_floor(sample_pos_t<float, UNIT>(czi, ds, un[2 * nv + cs])));
    }
    GWLoader<NT> ld;
    ld.plan(o, g);
    ld.issue(vn, plane, gwin);

    // byte offset of voxel e in a plane: off0 + e * estep where the voxel exists (x0 + lxb + XS e < nx, row in the grid)
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
                uu[d][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(make_rsrc(un + (size_t)d * nv, plane), voff(e), 0, LAGO_COMPOSE_AUX_LD));
        }
#pragma unroll
        for (int e = 0; e < U; ++e) {
            const int i = tl.x0 + lxb + XS * e;
            const bool in = L[e].setup(sample_pos_t<float, UNIT>(i, ds, uu[0][e]), sample_pos_t<float, UNIT>(j, ds, uu[1][e]),
                                       sample_pos_t<float, UNIT>(k, ds, uu[2][e]), g, o);
            outm |= (in | (voff(e) == GW::kOutside)) ? 0u : 1u << e;
            __builtin_amdgcn_sched_barrier(0);  // one sample at a time: interleaved, the eight setups need > 128 VGPRs
        }
    }
    // Three channels through the window.  STRAY: some sample of the workgroup has a corner outside the window; those
    // lanes take their corners with the pair gathers of Lerp3 instead (same value expression), wave by wave.
    auto channels = [&](auto stray) {
        constexpr bool STRAY = decltype(stray)::value;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (c > 0) {
                // channel c's window must have landed before anyone reads it: the wait for this wave's own LDS-direct
                // loads is spelled out (no language rule makes hipcc put it in front of the barrier; it does today)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
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
                            P.setup(sample_pos_t<float, UNIT>(i, ds, uu[0][e]), sample_pos_t<float, UNIT>(j, ds, uu[1][e]),
                                    sample_pos_t<float, UNIT>(k, ds, uu[2][e]), g.nx, g.ny, g.nz);
                            val = P.value(vn + (size_t)c * nv);
                        }
                    }
                }
                const float a = dsr * uu[c][e];
                const float b = dtr * val;
                // stored at once: holding the eight results back until the next window's loads are issued (one wait
                // for both) cost four spilled values -- 134 MB of scratch writes per launch at 32 x 3 x 128^3, 11 % of
                // the kernel's time
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, a + b), ro, voff(e), 0, LAGO_COMPOSE_AUX_ST);
            }
            if (c < 2) {
                __syncthreads();  // everyone has read channel c: the window is free
                ld.issue(vn + (size_t)(c + 1) * nv, plane, gwin);
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
template __global__ void compose3_window_kernel<GW::NT, 8, true>(float *, const float *, const float *, double, double, Geom, GWGrid);
template __global__ void compose3_window_kernel<GW::NT, 8, false>(float *, const float *, const float *, double, double, Geom, GWGrid);

std::atomic<int> g_gather_window{1};  // 1: LDS-window gathers where the shape allows (default); 0: pair gathers only
std::atomic<int> g_tile_cube{1};      // 1: 128^3 / 160^3 volumes take the instantiations with compile-time geometry

// Profiling builds only (-DLAGO_PROFILING; round 4, VERDICT r3 item 2, tools/ab_streams.py): LAGO_EXP_GATHER_PAD=<bytes>
// makes the gather kernels of the Euler step ask for at least that much LDS, so that only one of their workgroups fits a
// CU and the FFT passes of ANOTHER stream can co-reside (measured: 30 % slower, profiles/r04_stream_split.md).  The
// product library reads no environment variable here.
template <typename K>
static size_t padded_smem(K k, size_t smem) {
#ifdef LAGO_PROFILING
    static const size_t pad = [] {
        const char *e = getenv("LAGO_EXP_GATHER_PAD");
        return e ? (size_t)strtoul(e, nullptr, 10) : (size_t)0;
    }();
    if (pad > smem) {
        smem = pad;
        if (smem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    }
#else
    (void)k;
#endif
    return smem;
}

template <typename R>
static bool compose_window_launch(R *out, const R *u, const R *v, double ds, double dt, const Geom &g, int64_t nn,
                                  hipStream_t s) {
    if constexpr (sizeof(R) == 4) {
        GWGrid w;
        if (!make_gwgrid(w, g, nn) || ((uintptr_t)v & 15u)) return false;
        constexpr int NT = GW::NT, U = 8;
        const size_t smem = GW::lds_bytes<NT>();
        // (compiling the geometry of 128^3 / 160^3 volumes in, as ad_star3_tile_kernel does, costs this kernel 2-15
        // spilled registers at its 128: not done)
        if (unit_dt<R>(ds))
            hipLaunchKernelGGL((compose3_window_kernel<NT, U, true>), dim3(w.total), dim3(NT), padded_smem(compose3_window_kernel<NT, U, true>, smem), s, out, u, v, ds, dt, g, w);
        else
            hipLaunchKernelGGL((compose3_window_kernel<NT, U, false>), dim3(w.total), dim3(NT), padded_smem(compose3_window_kernel<NT, U, false>, smem), s, out, u, v, ds, dt, g, w);
        note_path(LP_GATHER_WINDOW);
        return true;
    }
    return false;
}

template <typename R>
static int compose_impl(R *out, const R *u, const R *v, double ds, double dt, int dim, int64_t nn, int64_t nx,
                        int64_t ny, int64_t nz, void *stream) {
    if (dim != 2 && dim != 3) return fail_invalid("Only two- and three-dimensional interpolation is supported");
    Geom g;
    if (!make_geom(g, dim, nn, nx, ny, nz)) return fail_invalid("compose: bad extent");
    if (g.nblocks == 0) return LAGO_OK;
    if (!out || !u || !v) return fail_invalid("compose: null pointer");
    hipStream_t s = (hipStream_t)stream;
    constexpr int U = 2;
    if (dim == 3 && g_interp_vec && g_gather_window && g.nvox >= 32768u && compose_window_launch(out, u, v, ds, dt, g, nn, s))
        return finish_launch(s, "compose");
    if (dim == 3 && g_interp_vec && g.nz >= 2 && kBlock / g.nz + 1 < g.ny && g.nvox >= 4u * U * kBlock) {
        const uint32_t nbx_u = (g.nvox + U * kBlock - 1) / (U * kBlock);
        const uint64_t nb = (uint64_t)nbx_u * (uint64_t)nn;
        if (nb < (1ull << 31)) {
            if (unit_dt<R>(ds))
                hipLaunchKernelGGL((compose3_unroll_kernel<R, U, true>), dim3((uint32_t)nb), dim3(kBlock), 0, s, out, u, v,
                                   ds, dt, g, nbx_u, (uint32_t)nb);
            else
                hipLaunchKernelGGL((compose3_unroll_kernel<R, U, false>), dim3((uint32_t)nb), dim3(kBlock), 0, s, out, u, v,
                                   ds, dt, g, nbx_u, (uint32_t)nb);
            note_path(LP_VECTOR_GATHER);
            return finish_launch(s, "compose");
        }
    }
    if (dim == 3)
        hipLaunchKernelGGL((compose_kernel<R, 3>), dim3(g.nblocks), dim3(kBlock), 0, s, out, u, v, ds, dt, g);
    else
        hipLaunchKernelGGL((compose_kernel<R, 2>), dim3(g.nblocks), dim3(kBlock), 0, s, out, u, v, ds, dt, g);
    return finish_launch(s, "compose");
}

// ------------------------------------------------------------------ Ad^*(phi, m) in one pass
//
// ad_star: out = (D phiinv + I) (m o (id + phiinv)), i.e. adjrep.Ad_star of the reference
// (/root/reference/lagomorph/adjrep.py:86-97): interp(m, phiinv) followed by
// jacobian_times_vectorfield(phiinv, ., displacement=True).  Unfused, the resampled momentum makes a
// round trip through HBM (12 + 12 bytes per voxel of 72); here it stays in registers.  The
// interpolated components are rounded to R exactly where the unfused sequence stores them, and the
// Jacobian product is the expression of jtv_fwd_kernel (diff.hip), so the result is bit-identical.

// 0.5 (f[+1] - f[-1]) with the neighbour offsets clamped at the borders (include/diff.h:55-76)
template <typename R>
__device__ __forceinline__ R cdiff(const R *__restrict__ f, int plus, int minus) {
    return (R)0.5f * (f[plus] - f[minus]);
}

template <typename R, int DIM>
__global__ __launch_bounds__(kBlock) void ad_star_kernel(R *__restrict__ out, R *__restrict__ mphi,
                                                         const R *__restrict__ phi, const R *__restrict__ m, Geom g) {
    const Vox vx = locate(g);
    if (!vx.valid) return;
    const size_t nv = g.nvox;
    const R *pn = phi + (size_t)vx.n * DIM * nv + vx.s;
    const R *mn = m + (size_t)vx.n * DIM * nv;
    R *on = out + (size_t)vx.n * DIM * nv + vx.s;
    R pv[DIM], wv[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) pv[d] = pn[(size_t)d * nv];
    int plus[DIM], minus[DIM];
    if (DIM == 3) {
        Lerp3<R> L;
        L.setup(sample_pos<R>(vx.i, 1.0, pv[0]), sample_pos<R>(vx.j, 1.0, pv[1]), sample_pos<R>(vx.k, 1.0, pv[2]),
                g.nx, g.ny, g.nz);
#pragma unroll
        for (int d = 0; d < 3; ++d) wv[d] = L.value(mn + (size_t)d * nv);
        const int P[3] = {vx.i, vx.j, vx.k}, Ln[3] = {g.nx, g.ny, g.nz}, St[3] = {g.ny * g.nz, g.nz, 1};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            plus[d] = P[d] + 1 < Ln[d] ? St[d] : 0;
            minus[d] = P[d] > 0 ? -St[d] : 0;
        }
    } else {
        Lerp2<R> L;
        L.setup(sample_pos<R>(vx.j, 1.0, pv[0]), sample_pos<R>(vx.k, 1.0, pv[1]), g.ny, g.nz);
#pragma unroll
        for (int d = 0; d < 2; ++d) wv[d] = L.value(mn + (size_t)d * nv);
        const int P[2] = {vx.j, vx.k}, Ln[2] = {g.ny, g.nz}, St[2] = {g.nz, 1};
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            plus[d] = P[d] + 1 < Ln[d] ? St[d] : 0;
            minus[d] = P[d] > 0 ? -St[d] : 0;
        }
    }
    if (mphi) {  // the resampled momentum, kept for the backward pass (what interp_forward would have stored)
#pragma unroll
        for (int d = 0; d < DIM; ++d) mphi[(size_t)vx.n * DIM * nv + (size_t)d * nv + vx.s] = wv[d];
    }
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
        R gq[DIM];
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
            gq[d] = cdiff(pn + (size_t)c * nv, plus[d], minus[d]);
            if (c == d) gq[d] = gq[d] + (R)1.0;
        }
        R sacc = lg_fma(gq[0], wv[0], gq[1] * wv[1]);  // dotw of diff.hip
        if (DIM == 3) sacc = lg_fma(gq[2], wv[2], sacc);
        on[(size_t)c * nv] = sacc;
    }
}

// Unrolled 3D variant: U slabs of 256 consecutive voxels per workgroup, one voxel of each per lane.
template <typename R, int U>
__global__ __launch_bounds__(kBlock) void ad_star3_unroll_kernel(R *__restrict__ out, R *__restrict__ mphi,
                                                                 const R *__restrict__ phi, const R *__restrict__ m,
                                                                 Geom g, uint32_t nbx_u, uint32_t nblocks_u) {
    const uint32_t Lb = block_order(blockIdx.x, nblocks_u, g.rev);
    const uint32_t n = Lb / nbx_u;  // uniform: scalar division
    const uint32_t bx = Lb - n * nbx_u;
    const size_t nv = g.nvox;
    const R *pn = phi + (size_t)n * 3 * nv;
    const R *mn = m + (size_t)n * 3 * nv;
    R *on = out + (size_t)n * 3 * nv;
    uint32_t s[U];
    bool ok[U];
    R pv[3][U];
#pragma unroll
    for (int e = 0; e < U; ++e) {
        s[e] = (bx * U + e) * kBlock + threadIdx.x;
        ok[e] = s[e] < g.nvox;
        if (!ok[e]) s[e] = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d) pv[d][e] = pn[(size_t)d * nv + s[e]];
    }
    Lerp3<R, false> L[U];  // nz >= 2 guaranteed by the host
    int plus[3][U], minus[3][U];
    uint32_t ci = 0, cj = 0, ck = 0;
    const uint32_t qj = (uint32_t)kBlock / (uint32_t)g.nz, rk = (uint32_t)kBlock % (uint32_t)g.nz;  // uniform
    const int syz = g.ny * g.nz;
#pragma unroll
    for (int e = 0; e < U; ++e) {
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
        // past-the-end lanes were redirected to voxel 0 for their loads; give them its coordinates
        const int i = ok[e] ? (int)ci : 0, j = ok[e] ? (int)cj : 0, k = ok[e] ? (int)ck : 0;
        L[e].setup(sample_pos_t<R, true>(i, 1.0, pv[0][e]), sample_pos_t<R, true>(j, 1.0, pv[1][e]),
                   sample_pos_t<R, true>(k, 1.0, pv[2][e]), g.nx, g.ny, g.nz);
        plus[0][e] = i + 1 < g.nx ? syz : 0;
        minus[0][e] = i > 0 ? -syz : 0;
        plus[1][e] = j + 1 < g.ny ? g.nz : 0;
        minus[1][e] = j > 0 ? -g.nz : 0;
        plus[2][e] = k + 1 < g.nz ? 1 : 0;
        minus[2][e] = k > 0 ? -1 : 0;
    }
    R wv[3][U];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (d) __builtin_amdgcn_sched_barrier(0);  // 16 pair loads in flight at a time, as in interp / compose
#pragma unroll
        for (int e = 0; e < U; ++e) wv[d][e] = L[e].value(mn + (size_t)d * nv);
    }
    if (mphi) {  // the resampled momentum, kept for the backward pass (what interp_forward would have stored)
        R *mo = mphi + (size_t)n * 3 * nv;
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int e = 0; e < U; ++e)
                if (ok[e]) mo[(size_t)d * nv + s[e]] = wv[d][e];
    }
    // keep the gather phase and the three stencil phases apart: hoisting the 72 stencil loads above
    // the lerps doubled the register count (201 VGPRs, 2 waves/SIMD) for no gain in overlap
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (c) __builtin_amdgcn_sched_barrier(0);
        // neighbours through a buffer descriptor on the component plane: 32-bit byte offsets,
        // no 64-bit address pair per load
        const BufRsrc pc = make_rsrc(pn + (size_t)c * nv, (uint32_t)(nv * sizeof(R)));
        R fp[3][U], fm[3][U];
#pragma unroll
        for (int e = 0; e < U; ++e) {
            const uint32_t sb = s[e] * (uint32_t)sizeof(R);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                fp[d][e] = buf_load1<R>(pc, sb + (uint32_t)(plus[d][e] * (int)sizeof(R)));
                fm[d][e] = buf_load1<R>(pc, sb + (uint32_t)(minus[d][e] * (int)sizeof(R)));
            }
        }
#pragma unroll
        for (int e = 0; e < U; ++e) {
            R gq[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                gq[d] = (R)0.5f * (fp[d][e] - fm[d][e]);
                if (c == d) gq[d] = gq[d] + (R)1.0;
            }
            const R sacc = lg_fma(gq[2], wv[2][e], lg_fma(gq[0], wv[0][e], gq[1] * wv[1][e]));
            if (ok[e]) on[(size_t)c * nv + s[e]] = sacc;
        }
    }
}


// Row-tile variant (stencil_tile.hpp): the three phiinv planes of a TX x TY-row tile are staged in LDS with a
// one-voxel halo and the 18 stencil neighbours of a voxel come from there -- 24 gathered + 3 centre + 3 (2 TX + 2 TY)
// / (TX TY) halo dwords per voxel through the vector-memory path instead of 24 + 3 + 18 (31.5 instead of 45 for the
// 2 x 4-row tile of a 128-voxel row).  The momentum gathers are issued before the barrier, so their latency and the
// staging overlap.  Arithmetic and its order are those of ad_star3_unroll_kernel: same bits.
// CUBE > 0: the volume is CUBE^3 and the tile the one make_row_tile picks for it (checked by the host) -- the geometry
// is a compile-time constant, which removes most of the scalar instructions of the tile / halo-row decode (about as
// many scalar as vector instructions per wave in the generic form: both issue ports of a SIMD are equally loaded).
template <typename R, int NT, int U, int RI, int ZC, int CUBE = 0>
__global__ __launch_bounds__(NT) void ad_star3_tile_kernel(R *__restrict__ out, R *__restrict__ mphi,
                                                           const R *__restrict__ phi, const R *__restrict__ m, Geom g,
                                                           RowTile t) {
    extern __shared__ __align__(16) unsigned char lago_smem[];
    R *lds = reinterpret_cast<R *>(lago_smem);
    if constexpr (CUBE > 0) {
        constexpr int rows = NT * U / CUBE, TY = rows / 2;   // 2 x TY rows (2 x 4 at 128, 2 x 3 at 160)
        g.nx = g.ny = g.nz = CUBE;
        g.nvox = (uint32_t)CUBE * CUBE * CUBE;
        t.TX = 2; t.TY = TY; t.RY = TY + 2; t.P = CUBE + 2;
        t.nhrows = 2u * TY + 4u;
        t.plane = 4u * (TY + 2) * (CUBE + 2);
        t.ntx = CUBE / 2; t.nty = (CUBE + TY - 1) / TY;
        t.tiles_per_item = t.ntx * t.nty;
        t.tile_vox = 2u * TY * CUBE;
        t.d_tiles = FastDiv(t.tiles_per_item); t.d_nty = FastDiv(t.nty); t.d_nz = FastDiv((uint32_t)CUBE);
        t.d_TY = FastDiv((uint32_t)TY); t.d_nhrows = FastDiv(t.nhrows);
    }
    const uint32_t Lb = block_order(blockIdx.x, t.total, g.rev);
    const uint32_t n = t.d_tiles.div(Lb);
    const uint32_t r = Lb - n * t.tiles_per_item;
    const uint32_t tx = t.d_nty.div(r), ty = r - tx * t.nty;
    const int x0 = (int)tx * t.TX, y0 = (int)ty * t.TY;
    const size_t nv = g.nvox;
    const R *pn = phi + (size_t)n * 3 * nv;
    const R *mn = m + (size_t)n * 3 * nv;
    R *on = out + (size_t)n * 3 * nv;
    TileVox q[U];
    R pv[3][U];
#pragma unroll
    for (int e = 0; e < U; ++e) {
        q[e] = tile_voxel(t, g, x0, y0, threadIdx.x + (uint32_t)e * NT);
#pragma unroll
        for (int d = 0; d < 3; ++d) pv[d][e] = ld_pol<LAGO_NT_AD_LD>(pn + (size_t)d * nv + q[e].s);
    }
    TileHalo<R, NT, 3, RI, ZC> halo;
    halo.issue(pn, nv, t, g, x0, y0);
    halo.commit(lds, g);
#pragma unroll
    for (int e = 0; e < U; ++e)
#pragma unroll
        for (int d = 0; d < 3; ++d) tile_put(lds + (size_t)d * t.plane, q[e], g.nz, pv[d][e]);
    Lerp3<R, false> L[U];  // nz >= 2 guaranteed by the host
#pragma unroll
    for (int e = 0; e < U; ++e)
        L[e].setup(sample_pos_t<R, true>(q[e].i, 1.0, pv[0][e]), sample_pos_t<R, true>(q[e].j, 1.0, pv[1][e]),
                   sample_pos_t<R, true>(q[e].k, 1.0, pv[2][e]), g.nx, g.ny, g.nz);
    R wv[3][U];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (d) __builtin_amdgcn_sched_barrier(0);  // 4 U pair loads in flight at a time, as in interp / compose
#pragma unroll
        for (int e = 0; e < U; ++e) wv[d][e] = L[e].value(mn + (size_t)d * nv);
    }
    if (mphi) {  // the resampled momentum, kept for the backward pass (what interp_forward would have stored)
        R *mo = mphi + (size_t)n * 3 * nv;
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int e = 0; e < U; ++e)
                if (q[e].ok) st_pol<LAGO_NT_AD_ST>(&mo[(size_t)d * nv + q[e].s], wv[d][e]);
    }
    __syncthreads();
    const uint32_t sy = (uint32_t)t.P, sx = (uint32_t)t.RY * (uint32_t)t.P;
    // all 18 U neighbour reads first (every lane has a valid slot index, so they need no predicate), then the products
    R fp[3][U][3], fm[3][U][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const R *pl = lds + (size_t)c * t.plane;
#pragma unroll
        for (int e = 0; e < U; ++e) {
            const uint32_t li = q[e].li;
            fp[c][e][0] = pl[li + sx]; fm[c][e][0] = pl[li - sx];
            fp[c][e][1] = pl[li + sy]; fm[c][e][1] = pl[li - sy];
            fp[c][e][2] = pl[li + 1];  fm[c][e][2] = pl[li - 1];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        R sacc[U];
#pragma unroll
        for (int e = 0; e < U; ++e) {
            R gq[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                gq[d] = (R)0.5f * (fp[c][e][d] - fm[c][e][d]);
                if (c == d) gq[d] = gq[d] + (R)1.0;
            }
            sacc[e] = lg_fma(gq[2], wv[2][e], lg_fma(gq[0], wv[0][e], gq[1] * wv[1][e]));
        }
#pragma unroll
        for (int e = 0; e < U; ++e)
            if (q[e].ok) {
                st_pol<LAGO_NT_AD_ST>(&on[(size_t)c * nv + q[e].s], sacc[e]);
            }
    }
}

std::atomic<int> g_stencil_tile{1};  // 1: LDS row-tile stencil kernels where the shape allows (default); 0: direct kernels

// 512 threads x 2 voxels: 8 rows of 128 voxels (2 x 4) or 6 rows of 160 (2 x 3) per workgroup, 63 VGPRs (four
// workgroups per CU).  Measured against 256 x 2 (2 x 2 rows), 1024 x 2 (4 x 4), 768 x 2, 256 x 4 and 512 x 4 voxels
// per thread, halo rows by LDS-direct loads, streaming stores and an earlier issue of the first gathers: all within
// +-3 % or slower (profiles/r03_stencil_tile.md).
template <typename R>
static bool ad_star_tile_launch(R *out, R *mphi, const R *phi, const R *m, const Geom &g, int64_t nn, hipStream_t s) {
    constexpr int NT = 512, U = 2, RI = 5;
    RowTile t;
    size_t smem;
    if (!make_row_tile(t, g, nn, NT * U, NT, 3, (int)sizeof(R), RI, smem)) return false;
    const int zc = (g.nz + 63) / 64;
#define LAGO_ADT(ZC)                                                                                               \
    hipLaunchKernelGGL((ad_star3_tile_kernel<R, NT, U, RI, ZC>), dim3(t.total), dim3(NT), smem, s, out, mphi, phi, m, g, t)
    if constexpr (sizeof(R) == 4) {
        // the two benchmark volumes with their geometry compiled in
        const bool cube = g.nx == g.ny && g.ny == g.nz && t.TX == 2 && g_tile_cube;
        if (cube && g.nz == 128 && t.TY == 4) {
            hipLaunchKernelGGL((ad_star3_tile_kernel<R, NT, U, RI, 2, 128>), dim3(t.total), dim3(NT), padded_smem(ad_star3_tile_kernel<R, NT, U, RI, 2, 128>, smem), s, out, mphi, phi, m, g, t);
            note_path(LP_STENCIL_TILE);
            return true;
        }
        if (cube && g.nz == 160 && t.TY == 3) {
            hipLaunchKernelGGL((ad_star3_tile_kernel<R, NT, U, RI, 3, 160>), dim3(t.total), dim3(NT), smem, s, out, mphi, phi, m, g, t);
            note_path(LP_STENCIL_TILE);
            return true;
        }
    }
    if (zc == 1) LAGO_ADT(1);
    else if (zc == 2) LAGO_ADT(2);
    else if (zc == 3) LAGO_ADT(3);
    else if (zc == 4) LAGO_ADT(4);
    else return false;
#undef LAGO_ADT
    note_path(LP_STENCIL_TILE);
    return true;
}

template <typename R>
static int ad_star_impl(R *out, R *mphi, const R *phi, const R *m, int dim, int64_t nn, int64_t nx, int64_t ny,
                        int64_t nz, void *stream) {
    if (dim != 2 && dim != 3) return fail_invalid("Only two- and three-dimensional fields are supported");
    Geom g;
    if (!make_geom(g, dim, nn, nx, ny, nz)) return fail_invalid("ad_star: bad extent");
    if (g.nblocks == 0) return LAGO_OK;
    if (!out || !phi || !m) return fail_invalid("ad_star: null pointer");
    if (out == phi || out == m || (mphi && (mphi == phi || mphi == m || mphi == out)))
        return fail_invalid("ad_star: outputs may not alias an input or each other");
    if (nx <= 1 || ny <= 1 || (dim == 3 && nz <= 1))
        return fail_invalid("Jacobian times vectorfield not implemented for 'thin' dimensions");
    hipStream_t s = (hipStream_t)stream;
    constexpr int U = 2;
    if (dim == 3 && g_interp_vec && g_stencil_tile && g.nvox >= 4096u && ad_star_tile_launch<R>(out, mphi, phi, m, g, nn, s))
        return finish_launch(s, "ad_star");
    if (dim == 3 && g_interp_vec && g.nz >= 2 && kBlock / g.nz + 1 < g.ny && g.nvox >= 4u * U * kBlock) {
        const uint32_t nbx_u = (g.nvox + U * kBlock - 1) / (U * kBlock);
        const uint64_t nb = (uint64_t)nbx_u * (uint64_t)nn;
        if (nb < (1ull << 31)) {
            hipLaunchKernelGGL((ad_star3_unroll_kernel<R, U>), dim3((uint32_t)nb), dim3(kBlock), 0, s, out, mphi, phi, m, g,
                               nbx_u, (uint32_t)nb);
            note_path(LP_VECTOR_GATHER);
            return finish_launch(s, "ad_star");
        }
    }
    if (dim == 3)
        hipLaunchKernelGGL((ad_star_kernel<R, 3>), dim3(g.nblocks), dim3(kBlock), 0, s, out, mphi, phi, m, g);
    else
        hipLaunchKernelGGL((ad_star_kernel<R, 2>), dim3(g.nblocks), dim3(kBlock), 0, s, out, mphi, phi, m, g);
    return finish_launch(s, "ad_star");
}

// out[i] = c0 x0[i] (+ c1 x1[i] (+ c2 x2[i] (+ c3 x3[i]))): the elementwise sums of the atlas step (gradient of the
// regulariser onto the velocity gradient; momentum update) in one pass each.  Evaluated left to right with one fma
// per term, coefficients rounded to R first -- the arithmetic of torch's add(alpha=...) chain on the same operands.
template <typename R, int K, int VEC>
__global__ __launch_bounds__(256) void lincomb_kernel(R *out, const R *x0, const R *x1, const R *x2, const R *x3, R c0,
                                                      R c1, R c2, R c3, size_t nvec, int rev) {
    struct alignas(sizeof(R) * VEC) V { R e[VEC]; };
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < nvec; j += stride) {
        const size_t i = rev ? nvec - 1 - j : j;   // launch direction (common.hpp)
        const V a = reinterpret_cast<const V *>(x0)[i];
        V b = a, c = a, d = a, r;
        if (K > 1) b = reinterpret_cast<const V *>(x1)[i];
        if (K > 2) c = reinterpret_cast<const V *>(x2)[i];
        if (K > 3) d = reinterpret_cast<const V *>(x3)[i];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            R acc = c0 * a.e[e];
            if (K > 1) acc = lg_fma(c1, b.e[e], acc);
            if (K > 2) acc = lg_fma(c2, c.e[e], acc);
            if (K > 3) acc = lg_fma(c3, d.e[e], acc);
            r.e[e] = acc;
        }
        reinterpret_cast<V *>(out)[i] = r;
    }
}

template <typename R>
int lincomb_impl(R *out, int k, const R *x0, const R *x1, const R *x2, const R *x3, double c0, double c1, double c2,
                 double c3, int64_t n, void *stream) {
    if (k < 1 || k > 4 || n < 0) return fail_invalid("lincomb: 1 to 4 terms");
    if (n == 0) return LAGO_OK;
    hipStream_t s = (hipStream_t)stream;
    constexpr int VEC = 16 / sizeof(R);
    auto aligned = [](const void *p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool vec = n % VEC == 0 && aligned(out) && aligned(x0) && aligned(k > 1 ? x1 : nullptr) &&
                     aligned(k > 2 ? x2 : nullptr) && aligned(k > 3 ? x3 : nullptr);
    const size_t nvec = vec ? (size_t)n / VEC : (size_t)n;
    const uint32_t grid = (uint32_t)std::min<size_t>((nvec + 255) / 256, (size_t)256 * 32);
    const int rev = next_direction();
#define LAGO_LC(K, V)                                                                                              \
    hipLaunchKernelGGL((lincomb_kernel<R, K, V>), dim3(grid), dim3(256), 0, s, out, x0, x1, x2, x3, (R)c0, (R)c1, (R)c2, \
                       (R)c3, nvec, rev)
    if (vec) {
        if (k == 1) LAGO_LC(1, VEC); else if (k == 2) LAGO_LC(2, VEC); else if (k == 3) LAGO_LC(3, VEC); else LAGO_LC(4, VEC);
    } else {
        if (k == 1) LAGO_LC(1, 1); else if (k == 2) LAGO_LC(2, 1); else if (k == 3) LAGO_LC(3, 1); else LAGO_LC(4, 1);
    }
#undef LAGO_LC
    return finish_launch(s, "lincomb");
}

}  // namespace lago

namespace lago {
void tune_fused(int stencil_tile, int gather_window) {
    // stencil_tile 0: direct kernels; 1: row tiles (default); 3: row tiles without the compile-time-geometry instantiations
    g_stencil_tile = stencil_tile ? 1 : 0;
    g_tile_cube = stencil_tile == 3 ? 0 : 1;
    g_gather_window = gather_window ? 1 : 0;
}
}  // namespace lago

extern "C" {
int lago_lincomb_f32(float *out, int k, const float *x0, const float *x1, const float *x2, const float *x3, double c0,
                     double c1, double c2, double c3, int64_t n, void *stream) {
    return lago::lincomb_impl<float>(out, k, x0, x1, x2, x3, c0, c1, c2, c3, n, stream);
}
int lago_lincomb_f64(double *out, int k, const double *x0, const double *x1, const double *x2, const double *x3,
                     double c0, double c1, double c2, double c3, int64_t n, void *stream) {
    return lago::lincomb_impl<double>(out, k, x0, x1, x2, x3, c0, c1, c2, c3, n, stream);
}
int lago_compose_f32(float *out, const float *u, const float *v, double ds, double dt, int dim, int64_t nn,
                     int64_t nx, int64_t ny, int64_t nz, void *stream) {
    return lago::compose_impl<float>(out, u, v, ds, dt, dim, nn, nx, ny, nz, stream);
}
int lago_compose_f64(double *out, const double *u, const double *v, double ds, double dt, int dim, int64_t nn,
                     int64_t nx, int64_t ny, int64_t nz, void *stream) {
    return lago::compose_impl<double>(out, u, v, ds, dt, dim, nn, nx, ny, nz, stream);
}
int lago_Ad_star_f32(float *out, float *mphi, const float *phiinv, const float *m, int dim, int64_t nn, int64_t nx,
                     int64_t ny, int64_t nz, void *stream) {
    return lago::ad_star_impl<float>(out, mphi, phiinv, m, dim, nn, nx, ny, nz, stream);
}
int lago_Ad_star_f64(double *out, double *mphi, const double *phiinv, const double *m, int dim, int64_t nn, int64_t nx,
                     int64_t ny, int64_t nz, void *stream) {
    return lago::ad_star_impl<double>(out, mphi, phiinv, m, dim, nn, nx, ny, nz, stream);
}
}
