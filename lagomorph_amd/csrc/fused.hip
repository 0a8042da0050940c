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

// 16-byte vectorised 3D variant (see interp_fwd3_vec_kernel in interp.hip): VPL consecutive-z
// voxels per lane; `gq` is the geometry of the VPL-groups.
template <typename R, int N>
struct alignas(sizeof(R) * N) VecN {
    R e[N];
};

template <typename R, int VPL>
__global__ __launch_bounds__(kBlock) void compose3_vec_kernel(R *__restrict__ out, const R *__restrict__ u,
                                                              const R *__restrict__ v, double ds, double dt, Geom gq) {
    typedef VecN<R, VPL> V;
    const Vox vx = locate(gq);
    if (!vx.valid) return;
    const int nz = gq.nz * VPL;
    const size_t nv = (size_t)gq.nvox * VPL;
    const size_t s = (size_t)vx.s * VPL;
    const R *un = u + (size_t)vx.n * 3 * nv + s;
    const R *vn = v + (size_t)vx.n * 3 * nv;
    R *on = out + (size_t)vx.n * 3 * nv + s;
    const R dsr = (R)ds, dtr = (R)dt;
    V uu[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) uu[d] = *reinterpret_cast<const V *>(un + (size_t)d * nv);
    Lerp3<R, false> L[VPL];
#pragma unroll
    for (int e = 0; e < VPL; ++e)
        L[e].setup(sample_pos<R>(vx.i, ds, uu[0].e[e]), sample_pos<R>(vx.j, ds, uu[1].e[e]),
                   sample_pos<R>(vx.k * VPL + e, ds, uu[2].e[e]), gq.nx, gq.ny, nz);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        V o;
#pragma unroll
        for (int e = 0; e < VPL; ++e) {
            const R a = dsr * uu[c].e[e];
            const R b = dtr * L[e].value(vn + (size_t)c * nv);
            o.e[e] = a + b;
        }
        *reinterpret_cast<V *>(on + (size_t)c * nv) = o;
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
    constexpr int VPL = 16 / sizeof(R);
    Geom gq;
    if (dim == 3 && g_interp_vec && nz % VPL == 0 && nz >= 2 * VPL &&
        (((uintptr_t)out | (uintptr_t)u | (uintptr_t)v) & 15) == 0 && make_geom(gq, 3, nn, nx, ny, nz / VPL)) {
        hipLaunchKernelGGL((compose3_vec_kernel<R, VPL>), dim3(gq.nblocks), dim3(kBlock), 0, s, out, u, v, ds, dt, gq);
        return finish_launch(s, "compose");
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
