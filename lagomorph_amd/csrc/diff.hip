// Finite-difference Jacobian-times-vector-field family -- gfx950 HIP kernels.
//
// Replaces cuda/diff.cu of the reference: forward (:17-127), backward
// (:187-473), adjoint forward (:546-632) and adjoint backward (:674-780).
// Central differences with index clamp (include/diff.h:7-76 over
// get_value_safe<CLAMP>, include/extrap.h:110-192), i.e. a one-sided half
// difference at the borders; the adjoint stencils are the reference's explicit
// three-case border formulas.
//
// One lane per voxel of the flattened (i, j, k) index, last axis fastest: the
// centre row of every operand streams as 256-byte wavefront rows, the +-1 row
// and +-1 slab neighbours are re-reads served by L1 / the XCD's L2 (workgroups
// of one XCD walk a contiguous range, common.hpp).
#include "common.hpp"

// Output stores of the stencil kernels (common.hpp: st_pol; profiles/r04_cache_policy.md): non-temporal for the two
// backward kernels (jtv_bwd -6 %, jtv_adj_bwd -3 %: their outputs no longer displace the +-1 slab neighbours in the
// L2), plain for the forward ones (jtv_adj_fwd measured 4 % slower with it)
#ifndef LAGO_NT_STENCIL_BWD_ST
#define LAGO_NT_STENCIL_BWD_ST 1
#endif
// jtv_adj_bwd reads v and w at the centre voxel only (once): non-temporal loads -9.5 % (185 -> 167 us at 8 x 3 x 128^3);
// the same for w in jtv_fwd: nothing
#ifndef LAGO_NT_ADJB_LD
#define LAGO_NT_ADJB_LD 1
#endif
#ifndef LAGO_NT_STENCIL_FWD_ST
#define LAGO_NT_STENCIL_FWD_ST 0
#endif

namespace lago {

// Signed element offsets of the clamped +-1 neighbours along the DIM axes, the
// position/extent on each axis and the plain axis strides.
template <int DIM>
struct Stencil {
    int plus[DIM], minus[DIM];  // clamped: 0 at the respective border
    int stride[DIM];
    int pos[DIM], len[DIM];
    __device__ __forceinline__ Stencil(const Geom &g, const Vox &v) {
        const int P[3] = {v.i, v.j, v.k};
        const int Ln[3] = {g.nx, g.ny, g.nz};
        const int St[3] = {g.ny * g.nz, g.nz, 1};
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
            const int a = d + 3 - DIM;  // 2D fields live on geometry axes (y, z)
            pos[d] = P[a];
            len[d] = Ln[a];
            stride[d] = St[a];
            plus[d] = P[a] + 1 < Ln[a] ? St[a] : 0;
            minus[d] = P[a] > 0 ? -St[a] : 0;
        }
    }
    // include/diff.h:55-76 grad_point
    template <typename R>
    __device__ __forceinline__ void grad(const R *__restrict__ f, R *gq) const {
#pragma unroll
        for (int d = 0; d < DIM; ++d) gq[d] = (R)0.5f * (f[plus[d]] - f[minus[d]]);
    }
    // Adjoint of the clamped central difference along axis d applied to a*b
    // (cuda/diff.cu:224-248, 334-391, 560-573, 603-620).  a, b point at the centre voxel.
    template <typename R>
    __device__ __forceinline__ R dT(const R *__restrict__ a, const R *__restrict__ b, int d) const {
        const int st = stride[d];
        if (pos[d] == 0) return (R)(-.5) * lg_fma(a[0], b[0], a[st] * b[st]);
        if (pos[d] == len[d] - 1) return (R)(.5) * lg_fma(a[0], b[0], a[-st] * b[-st]);
        return (R)(-.5) * lg_fma(a[st], b[st], -(a[-st] * b[-st]));
    }
    // Same formula on values that are already in registers (x0 = centre, xp / xm = the clamped +1 / -1
    // neighbours along axis d; the clamped side is never used at its border).  The kernels below load
    // every needed neighbour once up front: all loads are then independent and in flight together,
    // instead of up to six dependent-on-branch loads per (component, axis) pair.
    template <typename R>
    __device__ __forceinline__ R dTv(R a0, R ap, R am, R b0, R bp, R bm, int d) const {
        if (pos[d] == 0) return (R)(-.5) * lg_fma(a0, b0, ap * bp);
        if (pos[d] == len[d] - 1) return (R)(.5) * lg_fma(a0, b0, am * bm);
        return (R)(-.5) * lg_fma(ap, bp, -(am * bm));
    }
};

template <typename R, int DIM>
struct Nb {  // centre and clamped +-1 neighbours along every axis
    R c0, p[DIM], m[DIM];
    __device__ __forceinline__ void load(const Stencil<DIM> &st, const R *__restrict__ f) {
        c0 = f[0];
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
            p[d] = f[st.plus[d]];
            m[d] = f[st.minus[d]];
        }
    }
};

template <typename R, int DIM>
__device__ __forceinline__ R dotw(const R *g, const R *w) {
    R s = lg_fma(g[0], w[0], g[1] * w[1]);
    if (DIM == 3) s = lg_fma(g[2], w[2], s);
    return s;
}

// ------------------------------------------------------------------ (Dv + delta) w and its transpose

template <typename R, int DIM, bool DISP, bool TRANS>
__global__ __launch_bounds__(kBlock) void jtv_fwd_kernel(R *__restrict__ out, const R *__restrict__ v,
                                                         const R *__restrict__ w, int nc, Geom g) {
    const Vox vx = locate(g);
    if (!vx.valid) return;
    const size_t nv = g.nvox;
    const Stencil<DIM> st(g, vx);
    const R *vn = v + (size_t)vx.n * nc * nv + vx.s;
    const R *wn = w + (size_t)vx.n * DIM * nv + vx.s;
    R *on = out + (size_t)vx.n * nc * nv + vx.s;
    R wv[DIM], gq[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) wv[d] = wn[(size_t)d * nv];
    if (TRANS) {
        R acc[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            st.grad(vn + (size_t)c * nv, gq);
            if (DISP) gq[c] = gq[c] + (R)1.0;
#pragma unroll
            for (int d = 0; d < DIM; ++d) acc[d] = c == 0 ? gq[d] * wv[c] : lg_fma(gq[d], wv[c], acc[d]);
        }
#pragma unroll
        for (int d = 0; d < DIM; ++d) st_pol<LAGO_NT_STENCIL_FWD_ST>(&on[(size_t)d * nv], (R)(acc[d]));
    } else {
        for (int c = 0; c < nc; ++c) {
            st.grad(vn + (size_t)c * nv, gq);
            if (DISP) {
#pragma unroll
                for (int d = 0; d < DIM; ++d)
                    if (c == d) gq[d] = gq[d] + (R)1.0;
            }
            st_pol<LAGO_NT_STENCIL_FWD_ST>(&on[(size_t)c * nv], (R)(dotw<R, DIM>(gq, wv)));
        }
    }
}

template <typename R, int DIM, bool DISP, bool TRANS>
__global__ __launch_bounds__(kBlock) void jtv_bwd_kernel(R *__restrict__ d_v, R *__restrict__ d_w,
                                                         const R *__restrict__ go, const R *__restrict__ v,
                                                         const R *__restrict__ w, int nc, Geom g, int acc_v) {
    const Vox vx = locate(g);
    if (!vx.valid) return;
    const size_t nv = g.nvox;
    const Stencil<DIM> st(g, vx);
    const R *vn = v + (size_t)vx.n * nc * nv + vx.s;
    const R *wn = w + (size_t)vx.n * DIM * nv + vx.s;
    const R *gon = go + (size_t)vx.n * nc * nv + vx.s;
    R *dvn = d_v + (size_t)vx.n * nc * nv + vx.s;
    R *dwn = d_w + (size_t)vx.n * DIM * nv + vx.s;
    R gq[DIM];
    if (TRANS) {
        R gov[DIM];
#pragma unroll
        for (int d = 0; d < DIM; ++d) gov[d] = gon[(size_t)d * nv];
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            st.grad(vn + (size_t)c * nv, gq);
            if (DISP) gq[c] = gq[c] + (R)1.0;
            st_pol<LAGO_NT_STENCIL_BWD_ST>(&dwn[(size_t)c * nv], (R)((R)0 + dotw<R, DIM>(gq, gov)));
        }
        Nb<R, DIM> W[DIM], G[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            W[c].load(st, wn + (size_t)c * nv);
            G[c].load(st, gon + (size_t)c * nv);
        }
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            R acc = 0;
#pragma unroll
            for (int d = 0; d < DIM; ++d)
                acc = acc + st.dTv(W[c].c0, W[c].p[d], W[c].m[d], G[d].c0, G[d].p[d], G[d].m[d], d);
            st_pol<LAGO_NT_STENCIL_BWD_ST>(&dvn[(size_t)c * nv], (R)(acc_v ? dvn[(size_t)c * nv] + acc : acc));
        }
    } else {
        R dw[DIM];
        Nb<R, DIM> W[DIM];
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
            dw[d] = 0;
            W[d].load(st, wn + (size_t)d * nv);
        }
        for (int c = 0; c < nc; ++c) {
            st.grad(vn + (size_t)c * nv, gq);
            if (DISP) {
#pragma unroll
                for (int d = 0; d < DIM; ++d)
                    if (c == d) gq[d] = gq[d] + (R)1.0;
            }
            const R goc = gon[(size_t)c * nv];
#pragma unroll
            for (int d = 0; d < DIM; ++d) dw[d] = lg_fma(gq[d], goc, dw[d]);
            Nb<R, DIM> G;
            G.load(st, gon + (size_t)c * nv);
            R acc = 0;
#pragma unroll
            for (int d = 0; d < DIM; ++d) acc = acc + st.dTv(W[d].c0, W[d].p[d], W[d].m[d], G.c0, G.p[d], G.m[d], d);
            st_pol<LAGO_NT_STENCIL_BWD_ST>(&dvn[(size_t)c * nv], (R)(acc_v ? dvn[(size_t)c * nv] + acc : acc));
        }
#pragma unroll
        for (int d = 0; d < DIM; ++d) st_pol<LAGO_NT_STENCIL_BWD_ST>(&dwn[(size_t)d * nv], (R)(dw[d]));
    }
}

// ------------------------------------------------------------------ adjoint in the differentiated argument

template <typename R, int DIM>
__global__ __launch_bounds__(kBlock) void jtv_adj_fwd_kernel(R *__restrict__ out, const R *__restrict__ z,
                                                             const R *__restrict__ w, int nc, Geom g) {
    const Vox vx = locate(g);
    if (!vx.valid) return;
    const size_t nv = g.nvox;
    const Stencil<DIM> st(g, vx);
    const R *zn = z + (size_t)vx.n * nc * nv + vx.s;
    const R *wn = w + (size_t)vx.n * DIM * nv + vx.s;
    R *on = out + (size_t)vx.n * nc * nv + vx.s;
    Nb<R, DIM> W[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) W[d].load(st, wn + (size_t)d * nv);
    for (int c = 0; c < nc; ++c) {
        Nb<R, DIM> Z;
        Z.load(st, zn + (size_t)c * nv);
        R acc = 0;
#pragma unroll
        for (int d = 0; d < DIM; ++d) acc = acc + st.dTv(W[d].c0, W[d].p[d], W[d].m[d], Z.c0, Z.p[d], Z.m[d], d);
        st_pol<LAGO_NT_STENCIL_FWD_ST>(&on[(size_t)c * nv], (R)(acc));
    }
}

// ad^*(v, m) = (Dv)^T m - sum_d D_d^T (v_d m) in one pass: adjrep.ad_star (adjrep.py:69-83), which the
// reference evaluates as jacobian_times_vectorfield(v, m, transpose=True) minus
// jacobian_times_vectorfield_adjoint(m, v) -- two stencil kernels and a subtraction, 108 bytes per
// voxel; here 36.  Both terms are formed exactly as jtv_fwd_kernel<TRANS> and jtv_adj_fwd_kernel form
// them (each rounded to R), then subtracted: bit-identical to the three-call sequence.
template <typename R, int DIM>
__global__ __launch_bounds__(kBlock) void ad_star_small_kernel(R *__restrict__ out, const R *__restrict__ v,
                                                               const R *__restrict__ m, Geom g) {
    const Vox vx = locate(g);
    if (!vx.valid) return;
    const size_t nv = g.nvox;
    const Stencil<DIM> st(g, vx);
    const size_t base = (size_t)vx.n * DIM * nv + vx.s;
    const R *vn = v + base, *mn = m + base;
    R *on = out + base;
    Nb<R, DIM> V[DIM], M[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
        V[d].load(st, vn + (size_t)d * nv);
        M[d].load(st, mn + (size_t)d * nv);
    }
    // (Dv)^T m: acc[d] = sum_c (D_d v_c) m_c, accumulated over c as in jtv_fwd_kernel<TRANS>
    R a[DIM];
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
            const R gq = (R)0.5f * (V[c].p[d] - V[c].m[d]);
            a[d] = c == 0 ? gq * M[c].c0 : lg_fma(gq, M[c].c0, a[d]);
        }
    }
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
        R b = 0;
#pragma unroll
        for (int d = 0; d < DIM; ++d) b = b + st.dTv(V[d].c0, V[d].p[d], V[d].m[d], M[c].c0, M[c].p[d], M[c].m[d], d);
        st_pol<LAGO_NT_STENCIL_FWD_ST>(&on[(size_t)c * nv], (R)(a[c] - b));
    }
}

template <typename R, int DIM>
__global__ __launch_bounds__(kBlock) void jtv_adj_bwd_kernel(R *__restrict__ d_v, R *__restrict__ d_w,
                                                             const R *__restrict__ go, const R *__restrict__ v,
                                                             const R *__restrict__ w, Geom g) {
    const Vox vx = locate(g);
    if (!vx.valid) return;
    const size_t nv = g.nvox;
    const Stencil<DIM> st(g, vx);
    const size_t base = (size_t)vx.n * DIM * nv + vx.s;
    const R *vn = v + base, *wn = w + base, *gon = go + base;
    R *dvn = d_v + base, *dwn = d_w + base;
    R wv[DIM], dw[DIM], gq[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) wv[d] = ld_pol<LAGO_NT_ADJB_LD>(wn + (size_t)d * nv);
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
        st.grad(gon + (size_t)c * nv, gq);
        const R vc = ld_pol<LAGO_NT_ADJB_LD>(vn + (size_t)c * nv);
#pragma unroll
        for (int d = 0; d < DIM; ++d) dw[d] = c == 0 ? gq[d] * vc : lg_fma(gq[d], vc, dw[d]);
        st_pol<LAGO_NT_STENCIL_BWD_ST>(&dvn[(size_t)c * nv], (R)((R)0 + dotw<R, DIM>(gq, wv)));
    }
#pragma unroll
    for (int d = 0; d < DIM; ++d) st_pol<LAGO_NT_STENCIL_BWD_ST>(&dwn[(size_t)d * nv], (R)(dw[d]));
}

// ------------------------------------------------------------------ host entry points

static bool thin(int dim, int64_t nx, int64_t ny, int64_t nz) {
    return nx <= 1 || ny <= 1 || (dim == 3 && nz <= 1);
}

template <typename R>
static int jtv_forward_impl(R *out, const R *v, const R *w, int disp, int trans, int dim, int64_t nn, int64_t nc,
                            int64_t nx, int64_t ny, int64_t nz, void *stream) {
    if (dim != 2 && dim != 3)
        return fail_invalid("Only two- and three-dimensional jacobian times vectorfield is supported");
    if (thin(dim, nx, ny, nz)) return fail_invalid("Jacobian times vectorfield not implemented for 'thin' dimensions");
    if (disp && nc != dim) return fail_invalid("Displacement mode only defined for vector fields");
    if (trans && nc != dim) return fail_invalid("Jacobian transpose only implemented for vector fields");
    Geom g;
    if (nc < 0 || !make_geom(g, dim, nn, nx, ny, nz))
        return fail_invalid("jacobian_times_vectorfield_forward: bad extent");
    if (g.nblocks == 0 || nc == 0) return LAGO_OK;
    if (!out || !v || !w) return fail_invalid("jacobian_times_vectorfield_forward: null pointer");
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(D, DS, TR) \
    hipLaunchKernelGGL((jtv_fwd_kernel<R, D, DS, TR>), dim3(g.nblocks), dim3(kBlock), 0, s, out, v, w, (int)nc, g)
#define BY_FLAGS(D)                                        \
    do {                                                   \
        if (disp && trans) LAUNCH(D, true, true);          \
        else if (disp) LAUNCH(D, true, false);             \
        else if (trans) LAUNCH(D, false, true);            \
        else LAUNCH(D, false, false);                      \
    } while (0)
    if (dim == 3) BY_FLAGS(3); else BY_FLAGS(2);
#undef LAUNCH
    return finish_launch(s, "jacobian_times_vectorfield_forward");
}

template <typename R>
static int jtv_backward_impl(R *d_v, R *d_w, const R *go, const R *v, const R *w, int disp, int trans, int dim,
                             int64_t nn, int64_t nc, int64_t nx, int64_t ny, int64_t nz, void *stream, int acc_v = 0) {
    if (dim != 2 && dim != 3)
        return fail_invalid("Only two- and three-dimensional jacobian times vectorfield is supported");
    if (thin(dim, nx, ny, nz)) return fail_invalid("Jacobian times vectorfield not implemented for 'thin' dimensions");
    if (disp && nc != dim) return fail_invalid("Displacement mode only defined for vector fields");
    if (trans && nc != dim) return fail_invalid("Jacobian transpose only implemented for vector fields");
    Geom g;
    if (nc < 0 || !make_geom(g, dim, nn, nx, ny, nz))
        return fail_invalid("jacobian_times_vectorfield_backward: bad extent");
    hipStream_t s = (hipStream_t)stream;
    if (g.nblocks == 0) return LAGO_OK;
    if (!d_w || (nc && (!d_v || !go || !v || !w)))
        return fail_invalid("jacobian_times_vectorfield_backward: null pointer");
    if (nc == 0) {  // d_w is all zeros, d_v is empty
        LAGO_HIP_TRY(hipMemsetAsync(d_w, 0, (size_t)nn * dim * g.nvox * sizeof(R), s));
        return finish_launch(s, "jacobian_times_vectorfield_backward");
    }
#define LAUNCH(D, DS, TR)                                                                                       \
    hipLaunchKernelGGL((jtv_bwd_kernel<R, D, DS, TR>), dim3(g.nblocks), dim3(kBlock), 0, s, d_v, d_w, go, v, w, \
                       (int)nc, g, acc_v)
    if (dim == 3) BY_FLAGS(3); else BY_FLAGS(2);
#undef LAUNCH
#undef BY_FLAGS
    return finish_launch(s, "jacobian_times_vectorfield_backward");
}

template <typename R>
static int jtv_adjoint_forward_impl(R *out, const R *z, const R *w, int dim, int64_t nn, int64_t nc, int64_t nx,
                                    int64_t ny, int64_t nz, void *stream) {
    if (dim != 2 && dim != 3)
        return fail_invalid("Only two- and three-dimensional jacobian times vectorfield is supported");
    if (thin(dim, nx, ny, nz)) return fail_invalid("Jacobian times vectorfield not implemented for 'thin' dimensions");
    Geom g;
    if (nc < 0 || !make_geom(g, dim, nn, nx, ny, nz))
        return fail_invalid("jacobian_times_vectorfield_adjoint_forward: bad extent");
    if (g.nblocks == 0 || nc == 0) return LAGO_OK;
    if (!out || !z || !w) return fail_invalid("jacobian_times_vectorfield_adjoint_forward: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (dim == 3)
        hipLaunchKernelGGL((jtv_adj_fwd_kernel<R, 3>), dim3(g.nblocks), dim3(kBlock), 0, s, out, z, w, (int)nc, g);
    else
        hipLaunchKernelGGL((jtv_adj_fwd_kernel<R, 2>), dim3(g.nblocks), dim3(kBlock), 0, s, out, z, w, (int)nc, g);
    return finish_launch(s, "jacobian_times_vectorfield_adjoint_forward");
}

template <typename R>
static int ad_star_small_impl(R *out, const R *v, const R *m, int dim, int64_t nn, int64_t nx, int64_t ny, int64_t nz,
                              void *stream) {
    if (dim != 2 && dim != 3)
        return fail_invalid("Only two- and three-dimensional jacobian times vectorfield is supported");
    if (thin(dim, nx, ny, nz)) return fail_invalid("Jacobian times vectorfield not implemented for 'thin' dimensions");
    Geom g;
    if (!make_geom(g, dim, nn, nx, ny, nz)) return fail_invalid("ad_star: bad extent");
    if (g.nblocks == 0) return LAGO_OK;
    if (!out || !v || !m) return fail_invalid("ad_star: null pointer");
    if (out == v || out == m) return fail_invalid("ad_star: out may not alias an input");
    hipStream_t s = (hipStream_t)stream;
    if (dim == 3)
        hipLaunchKernelGGL((ad_star_small_kernel<R, 3>), dim3(g.nblocks), dim3(kBlock), 0, s, out, v, m, g);
    else
        hipLaunchKernelGGL((ad_star_small_kernel<R, 2>), dim3(g.nblocks), dim3(kBlock), 0, s, out, v, m, g);
    return finish_launch(s, "ad_star");
}

template <typename R>
static int jtv_adjoint_backward_impl(R *d_v, R *d_w, const R *go, const R *v, const R *w, int dim, int64_t nn,
                                     int64_t nx, int64_t ny, int64_t nz, void *stream) {
    if (dim != 2 && dim != 3)
        return fail_invalid("Only two- and three-dimensional jacobian times vectorfield is supported");
    if (thin(dim, nx, ny, nz)) return fail_invalid("Jacobian times vectorfield not implemented for 'thin' dimensions");
    Geom g;
    if (!make_geom(g, dim, nn, nx, ny, nz))
        return fail_invalid("jacobian_times_vectorfield_adjoint_backward: bad extent");
    if (g.nblocks == 0) return LAGO_OK;
    if (!d_v || !d_w || !go || !v || !w)
        return fail_invalid("jacobian_times_vectorfield_adjoint_backward: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (dim == 3)
        hipLaunchKernelGGL((jtv_adj_bwd_kernel<R, 3>), dim3(g.nblocks), dim3(kBlock), 0, s, d_v, d_w, go, v, w, g);
    else
        hipLaunchKernelGGL((jtv_adj_bwd_kernel<R, 2>), dim3(g.nblocks), dim3(kBlock), 0, s, d_v, d_w, go, v, w, g);
    return finish_launch(s, "jacobian_times_vectorfield_adjoint_backward");
}

}  // namespace lago

extern "C" {
#define LAGO_DEFINE(REAL, SUF)                                                                                      \
    int lago_jtv_forward##SUF(REAL *out, const REAL *v, const REAL *w, int displacement, int transpose, int dim,   \
                              int64_t nn, int64_t nc, int64_t nx, int64_t ny, int64_t nz, void *stream) {          \
        return lago::jtv_forward_impl<REAL>(out, v, w, displacement, transpose, dim, nn, nc, nx, ny, nz, stream);  \
    }                                                                                                               \
    int lago_jtv_backward##SUF(REAL *d_v, REAL *d_w, const REAL *go, const REAL *v, const REAL *w,                 \
                               int displacement, int transpose, int dim, int64_t nn, int64_t nc, int64_t nx,       \
                               int64_t ny, int64_t nz, void *stream) {                                             \
        return lago::jtv_backward_impl<REAL>(d_v, d_w, go, v, w, displacement, transpose, dim, nn, nc, nx, ny, nz, \
                                             stream);                                                              \
    }                                                                                                               \
    int lago_jtv_adjoint_forward##SUF(REAL *out, const REAL *z, const REAL *w, int dim, int64_t nn, int64_t nc,    \
                                      int64_t nx, int64_t ny, int64_t nz, void *stream) {                          \
        return lago::jtv_adjoint_forward_impl<REAL>(out, z, w, dim, nn, nc, nx, ny, nz, stream);                   \
    }                                                                                                               \
    int lago_jtv_backward_acc##SUF(REAL *d_v, REAL *d_w, const REAL *go, const REAL *v, const REAL *w,             \
                                   int displacement, int transpose, int dim, int64_t nn, int64_t nc, int64_t nx,   \
                                   int64_t ny, int64_t nz, int acc_v, void *stream) {                              \
        return lago::jtv_backward_impl<REAL>(d_v, d_w, go, v, w, displacement, transpose, dim, nn, nc, nx, ny, nz, \
                                             stream, acc_v);                                                       \
    }                                                                                                              \
    int lago_jtv_adjoint_backward##SUF(REAL *d_v, REAL *d_w, const REAL *go, const REAL *v, const REAL *w,         \
                                       int dim, int64_t nn, int64_t nx, int64_t ny, int64_t nz, void *stream) {    \
        return lago::jtv_adjoint_backward_impl<REAL>(d_v, d_w, go, v, w, dim, nn, nx, ny, nz, stream);             \
    }
LAGO_DEFINE(float, _f32)
LAGO_DEFINE(double, _f64)
#undef LAGO_DEFINE
int lago_ad_star_f32(float *out, const float *v, const float *m, int dim, int64_t nn, int64_t nx, int64_t ny,
                     int64_t nz, void *stream) {
    return lago::ad_star_small_impl<float>(out, v, m, dim, nn, nx, ny, nz, stream);
}
int lago_ad_star_f64(double *out, const double *v, const double *m, int dim, int64_t nn, int64_t nx, int64_t ny,
                     int64_t nz, void *stream) {
    return lago::ad_star_small_impl<double>(out, v, m, dim, nn, nx, ny, nz, stream);
}
}
