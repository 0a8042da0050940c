// Fused geometry operators -- gfx950 HIP kernels.
//
// compose: out = ds*u + dt*interp(v, u, ds), i.e. deform.compose of the reference
// (/root/reference/lagomorph/deform.py:53-55), which there is one interp kernel
// plus three elementwise torch kernels (84 extra bytes per voxel of traffic for a
// 3-vector field).  The three roundings of the unfused expression are kept
// (fl(fl(ds*u) + fl(dt*I))), so the result is bit-identical to evaluating the
// reference formula with this library's interp.
#include "common.hpp"

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
    const uint32_t Lb = xcd_swizzle(blockIdx.x, nblocks_u);
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

template <typename R>
static int compose_impl(R *out, const R *u, const R *v, double ds, double dt, int dim, int64_t nn, int64_t nx,
                        int64_t ny, int64_t nz, void *stream) {
    if (dim != 2 && dim != 3) return fail_invalid("Only two- and three-dimensional interpolation is supported");
    Geom g;
    if (!make_geom(g, dim, nn, nx, ny, nz)) return fail_invalid("compose: bad extent");
    if (g.nblocks == 0) return LAGO_OK;
    if (!out || !u || !v) return fail_invalid("compose: null pointer");
    hipStream_t s = (hipStream_t)stream;
    constexpr int U = 4;
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
            return finish_launch(s, "compose");
        }
    }
    if (dim == 3)
        hipLaunchKernelGGL((compose_kernel<R, 3>), dim3(g.nblocks), dim3(kBlock), 0, s, out, u, v, ds, dt, g);
    else
        hipLaunchKernelGGL((compose_kernel<R, 2>), dim3(g.nblocks), dim3(kBlock), 0, s, out, u, v, ds, dt, g);
    return finish_launch(s, "compose");
}

}  // namespace lago

extern "C" {
int lago_compose_f32(float *out, const float *u, const float *v, double ds, double dt, int dim, int64_t nn,
                     int64_t nx, int64_t ny, int64_t nz, void *stream) {
    return lago::compose_impl<float>(out, u, v, ds, dt, dim, nn, nx, ny, nz, stream);
}
int lago_compose_f64(double *out, const double *u, const double *v, double ds, double dt, int dim, int64_t nn,
                     int64_t nx, int64_t ny, int64_t nz, void *stream) {
    return lago::compose_impl<double>(out, u, v, ds, dt, dim, nn, nx, ny, nz, stream);
}
}
