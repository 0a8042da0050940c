// The fluid operator on ONE frequency bin (cuda/metric.cu:20-160, :220-306): the symmetric 3 x 3 (2 x 2) symbol squared,
// applied to the real and the imaginary parts of the three (two) components, or its inverse through a Cholesky factor.
// One source for metric.hip's operator kernel and for the FFT passes that apply the operator between their forward and
// inverse stages (fftg.hip): the same expressions in the same order, hence the same bits.
#pragma once
#include "common.hpp"

namespace lago {

template <typename R>
__device__ __forceinline__ R fb_safe_sqrt(R x) {
    if ((double)x < 1e-8) return (R)1e-4;
    return sizeof(R) == 4 ? (R)sqrtf((float)x) : (R)sqrt((double)x);  // correctly rounded (IEEE) forms
}
template <typename R>
__device__ __forceinline__ R fb_recip_via_double(R x) {  // `1./x` with x a Real: double division, narrowed
    return (R)(1. / (double)x);
}

template <typename R, bool INV>
struct FluidBin3 {
    R L00, L10, L11, L20, L21, L22;
    R ooG00, G10, ooG11, G20, G21, ooG22;
    __device__ __forceinline__ void setup(R wx, R wy, R wz, R sx, R sy, R sz, double alpha, double beta, double gamma) {
        const R lambda = (R)__builtin_fma(alpha, (double)(wx + wy + wz), gamma);
        const R l00 = (R)__builtin_fma(-beta, (double)wx, (double)lambda);
        const R l11 = (R)__builtin_fma(-beta, (double)wy, (double)lambda);
        const R l22 = (R)__builtin_fma(-beta, (double)wz, (double)lambda);
        const R l10 = (R)(beta * (double)sx * (double)sy);
        const R l20 = (R)(beta * (double)sx * (double)sz);
        const R l21 = (R)(beta * (double)sy * (double)sz);
        L00 = lg_fma(l20, l20, lg_fma(l00, l00, l10 * l10));
        L10 = lg_fma(l20, l21, lg_fma(l00, l10, l10 * l11));
        L11 = lg_fma(l21, l21, lg_fma(l10, l10, l11 * l11));
        L20 = lg_fma(l20, l22, lg_fma(l00, l20, l10 * l21));
        L21 = lg_fma(l21, l22, lg_fma(l10, l20, l11 * l21));
        L22 = lg_fma(l22, l22, lg_fma(l20, l20, l21 * l21));
        ooG00 = G10 = ooG11 = G20 = G21 = ooG22 = 0;
        if (INV) {  // cuda/metric.cu:47-78
            ooG00 = fb_recip_via_double(fb_safe_sqrt(L00));
            G10 = L10 * ooG00;
            G20 = L20 * ooG00;
            ooG11 = lg_fma(-G10, G10, L11);
            ooG11 = fb_recip_via_double(fb_safe_sqrt(ooG11));
            G21 = lg_fma(-G20, G10, L21) * ooG11;
            ooG22 = lg_fma(-G21, G21, lg_fma(-G20, G20, L22));
            ooG22 = fb_recip_via_double(fb_safe_sqrt(ooG22));
        }
    }
    __device__ __forceinline__ void apply(R &bX, R &bY, R &bZ) const {
        if (INV) {  // cuda/metric.cu:103-130
            R y0 = bX * ooG00;
            R y1 = lg_fma(-G10, y0, bY) * ooG11;
            R y2 = lg_fma(-G21, y1, lg_fma(-G20, y0, bZ)) * ooG22;
            bZ = y2 * ooG22;
            bY = lg_fma(-G21, bZ, y1) * ooG11;
            bX = lg_fma(-G20, bZ, lg_fma(-G10, bY, y0)) * ooG00;
        } else {  // cuda/metric.cu:145-160
            R x = lg_fma(L20, bZ, lg_fma(L00, bX, L10 * bY));
            R y = lg_fma(L21, bZ, lg_fma(L10, bX, L11 * bY));
            bZ = lg_fma(L22, bZ, lg_fma(L20, bX, L21 * bY));
            bX = x;
            bY = y;
        }
    }
};

// the 2D operator (cuda/metric.cu:20-45, :80-101, :132-143, :162-218)
template <typename R, bool INV>
struct FluidBin2 {
    R L00, L10, L11, ooG00, G10, ooG11;
    __device__ __forceinline__ void setup(R wx, R wy, R sx, R sy, double alpha, double beta, double gamma) {
        const R lambda = (R)__builtin_fma(alpha, (double)(wx + wy), gamma);
        const R l00 = (R)__builtin_fma(-beta, (double)wx, (double)lambda);
        const R l11 = (R)__builtin_fma(-beta, (double)wy, (double)lambda);
        const R l10 = (R)(beta * (double)sx * (double)sy);
        L00 = lg_fma(l00, l00, l10 * l10);
        L10 = lg_fma(l00, l10, l10 * l11);
        L11 = lg_fma(l11, l11, l10 * l10);
        ooG00 = G10 = ooG11 = 0;
        if (INV) {  // cuda/metric.cu:20-45
            ooG00 = fb_recip_via_double(fb_safe_sqrt(L00));
            G10 = L10 * ooG00;
            ooG11 = lg_fma(-G10, G10, L11);
            ooG11 = fb_recip_via_double(fb_safe_sqrt(ooG11));
        }
    }
    __device__ __forceinline__ void apply(R &bX, R &bY) const {
        if (INV) {  // cuda/metric.cu:80-101
            R y0 = bX * ooG00;
            R y1 = lg_fma(-G10, y0, bY) * ooG11;
            bY = y1 * ooG11;
            bX = lg_fma(-G10, bY, y0) * ooG00;
        } else {  // cuda/metric.cu:132-143
            R x = lg_fma(L00, bX, L10 * bY);
            bY = lg_fma(L10, bX, L11 * bY);
            bX = x;
        }
    }
};

}  // namespace lago
