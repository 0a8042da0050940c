// Fluid metric operator in the Fourier domain -- gfx950 HIP kernel.
//
// Replaces cuda/metric.cu of the reference (fluid_kernel_2d :162-218,
// fluid_kernel_3d :220-306, Cholesky helpers :20-130, OperatorMultiply
// :132-160).  Per frequency bin the symmetric matrix l (lambda on the diagonal,
// beta*sin*sin off it) is squared to L = l*l and either applied (flat) or
// inverted through its Cholesky factor (sharp), on the real and imaginary part
// of every vector component, in place on the interleaved-complex rFFT buffer.
//
// One lane per complex bin of the flattened (kx, ky, kz) index (kz fastest, 8 or
// 16 contiguous bytes per lane), batch loop inside the lane so the factor is
// computed once per bin exactly as the reference does.  Pure streaming: every
// byte of Fm is read once and written once.
#include "common.hpp"
#include "fluid_bin.hpp"

namespace lago {

template <typename R>
struct alignas(2 * sizeof(R)) Cplx {
    R re, im;
};

// cuda/metric.cu:14-18
template <typename R, int DIM, bool INV>
__global__ __launch_bounds__(kBlock) void fluid_kernel(Cplx<R> *__restrict__ Fm, const R *__restrict__ cosX,
                                                       const R *__restrict__ sinX, const R *__restrict__ cosY,
                                                       const R *__restrict__ sinY, const R *__restrict__ cosZ,
                                                       const R *__restrict__ sinZ, double alpha, double beta,
                                                       double gamma, int nn, Geom g, R scale) {
    const Vox v = locate(g);
    if (!v.valid) return;
    const size_t nv = g.nvox;  // complex bins per component
    Cplx<R> *F = Fm + v.s;
    if (DIM == 3) {
        FluidBin3<R, INV> op;   // (fluid_bin.hpp: shared with the FFT passes that apply the operator in place)
        op.setup(cosX[v.i], cosY[v.j], cosZ[v.k], sinX[v.i], sinY[v.j], sinZ[v.k], alpha, beta, gamma);
        for (int n = 0; n < nn; ++n, F += 3 * nv) {
            Cplx<R> a = F[0], b = F[nv], c = F[2 * nv];
            R X[2] = {a.re, a.im}, Y[2] = {b.re, b.im}, Z[2] = {c.re, c.im};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                R bX = X[q], bY = Y[q], bZ = Z[q];
                op.apply(bX, bY, bZ);
                X[q] = bX; Y[q] = bY; Z[q] = bZ;
            }
            F[0] = Cplx<R>{X[0] * scale, X[1] * scale};  // scale == 1 is a bitwise no-op
            F[nv] = Cplx<R>{Y[0] * scale, Y[1] * scale};
            F[2 * nv] = Cplx<R>{Z[0] * scale, Z[1] * scale};
        }
    } else {
        FluidBin2<R, INV> op;
        op.setup(cosX[v.j], cosY[v.k], sinX[v.j], sinY[v.k], alpha, beta, gamma);
        for (int n = 0; n < nn; ++n, F += 2 * nv) {
            Cplx<R> a = F[0], b = F[nv];
            R X[2] = {a.re, a.im}, Y[2] = {b.re, b.im};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                R bX = X[q], bY = Y[q];
                op.apply(bX, bY);
                X[q] = bX; Y[q] = bY;
            }
            F[0] = Cplx<R>{X[0] * scale, X[1] * scale};
            F[nv] = Cplx<R>{Y[0] * scale, Y[1] * scale};
        }
    }
}

template <typename R>
int fluid_operator_impl(R *Fm, int inverse, const R *cosX, const R *sinX, const R *cosY, const R *sinY,
                        const R *cosZ, const R *sinZ, double alpha, double beta, double gamma, int dim,
                        int64_t nn, int64_t nx, int64_t ny, int64_t nz, void *stream, double scale) {
    if (dim != 2 && dim != 3) return fail_invalid("Only two- and three-dimensional fluid metric is supported");
    Geom g;
    if (nn < 0 || nn >= (1ll << 31) || !make_geom(g, dim, 1, nx, ny, nz))
        return fail_invalid("fluid_operator: bad extent");
    if (g.nblocks == 0 || nn == 0) return LAGO_OK;
    if (!Fm || !cosX || !sinX || !cosY || !sinY || (dim == 3 && (!cosZ || !sinZ)))
        return fail_invalid("fluid_operator: null pointer");
    if (((uintptr_t)Fm) % (2 * sizeof(R))) return fail_invalid("fluid_operator: Fmv must be aligned to a complex element");
    hipStream_t s = (hipStream_t)stream;
    Cplx<R> *F = reinterpret_cast<Cplx<R> *>(Fm);
#define LAUNCH(D, INV)                                                                                           \
    hipLaunchKernelGGL((fluid_kernel<R, D, INV>), dim3(g.nblocks), dim3(kBlock), 0, s, F, cosX, sinX, cosY, sinY, \
                       cosZ, sinZ, alpha, beta, gamma, (int)nn, g, (R)scale)
    if (dim == 3) {
        if (inverse) LAUNCH(3, true); else LAUNCH(3, false);
    } else {
        if (inverse) LAUNCH(2, true); else LAUNCH(2, false);
    }
#undef LAUNCH
    return finish_launch(s, "fluid_operator");
}

template int fluid_operator_impl<float>(float *, int, const float *, const float *, const float *, const float *,
                                        const float *, const float *, double, double, double, int, int64_t, int64_t,
                                        int64_t, int64_t, void *, double);
template int fluid_operator_impl<double>(double *, int, const double *, const double *, const double *,
                                         const double *, const double *, const double *, double, double, double, int,
                                         int64_t, int64_t, int64_t, int64_t, void *, double);

}  // namespace lago

extern "C" {
#define LAGO_DEFINE(REAL, SUF)                                                                                    \
    int lago_fluid_operator##SUF(REAL *Fm, int inverse, const REAL *cosX, const REAL *sinX, const REAL *cosY,     \
                                 const REAL *sinY, const REAL *cosZ, const REAL *sinZ, double alpha, double beta, \
                                 double gamma, int dim, int64_t nn, int64_t nx, int64_t ny, int64_t nz,           \
                                 void *stream) {                                                                  \
        return lago::fluid_operator_impl<REAL>(Fm, inverse, cosX, sinX, cosY, sinY, cosZ, sinZ, alpha, beta,      \
                                               gamma, dim, nn, nx, ny, nz, stream, 1.0);                          \
    }
LAGO_DEFINE(float, _f32)
LAGO_DEFINE(double, _f64)
#undef LAGO_DEFINE
}
