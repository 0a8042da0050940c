// Fused x-axis pass of FluidMetric sharp/flat -- gfx950, float32, 3D, nx in {64, 128, 256}.
//
// rocFFT executes the 3D real transform as one pass along z and two strided "column" passes along
// y and x; with the per-frequency operator in between, sharp/flat makes seven passes over the
// half-spectrum (0.32 + 0.46 + 0.46 | 0.36 | 0.46 + 0.46 + 0.34 ms at 32 x 3 x 128^3, profile A/B),
// and the two strided column kernels move 1.7x their payload.  This kernel replaces three of the
// seven (forward x-FFT, operator, inverse x-FFT) by ONE pass: a workgroup loads the three vector
// components of a (y, kz-chunk) column bundle for all nx into LDS, runs a radix-2
// decimation-in-frequency FFT along x in place (natural -> bit-reversed order), applies the
// per-frequency operator with kx = bitrev(position) (so no reordering pass is needed), runs the
// decimation-in-time inverse (bit-reversed -> natural) and stores the bundle back in place.
// The (y, z) transforms stay with rocFFT as a batched 2D real plan.
//
// The operator coefficients (L^2 entries for flat, its Cholesky factor for sharp) depend only on
// the frequency, not on the batch item; the reference amortises them with a batch loop inside the
// thread (cuda/metric.cu:272-304).  Here they are tabulated once per (shape, parameters) by
// fluid_coef_kernel -- same expressions, same roundings as fluid_kernel in metric.hip -- and read
// back (24 bytes per bin, 3 % of the payload).
#include "common.hpp"

namespace lago {

// cuda/metric.cu:14-18
__device__ __forceinline__ float fx_safe_sqrt(float x) {
    if ((double)x < 1e-8) return (float)1e-4;
    return sqrtf(x);
}
__device__ __forceinline__ float fx_recip(float x) { return (float)(1. / (double)x); }

// frequency held at position p of a digit-reversed n-point spectrum, n = R * 2^L2 (fft_lds.hpp: freq_at<Sz<R, L2>>)
__device__ __forceinline__ int freq_at_rt(int p, int n) {
    const int l2 = __builtin_ctz((unsigned)n), r = n >> l2;
    const int low = p & ((1 << l2) - 1);
    const int br = l2 ? (int)(__builtin_bitreverse32((uint32_t)low) >> (32 - l2)) : 0;
    return r * br + (p >> l2);
}

// One lane per frequency bin (kx, ky, kz): coefficients of cuda/metric.cu:240-270 in the layout
// tab[bin][6] = {L00 L10 L11 L20 L21 L22} (flat) or {ooG00 G10 ooG11 G20 G21 ooG22} (sharp).
template <bool INV>
__global__ __launch_bounds__(kBlock) void fluid_coef_kernel(float *__restrict__ tab, const float *__restrict__ cosX,
                                                            const float *__restrict__ sinX, const float *__restrict__ cosY,
                                                            const float *__restrict__ sinY, const float *__restrict__ cosZ,
                                                            const float *__restrict__ sinZ, double alpha, double beta,
                                                            double gamma, Geom g, int split) {
    const Vox v = locate(g);
    if (!v.valid) return;
    // split layout: (v.j, v.k) are POSITIONS (r, q) of the zy passes' digit-reversed spectrum (fft_lds.hpp), the bin
    // they hold is (freq_at(r), freq_at(q)); the spare column q = nzh is the Nyquist bin
    const int nzh = g.nz - 1;
    const int ky = split ? freq_at_rt(v.j, g.ny) : v.j, kz = split && v.k < nzh ? freq_at_rt(v.k, nzh) : v.k;
    const float wx = cosX[v.i], wy = cosY[ky], wz = cosZ[kz];
    const float sx = sinX[v.i], sy = sinY[ky], sz = sinZ[kz];
    const float lambda = (float)__builtin_fma(alpha, (double)(wx + wy + wz), gamma);
    const float l00 = (float)__builtin_fma(-beta, (double)wx, (double)lambda);
    const float l11 = (float)__builtin_fma(-beta, (double)wy, (double)lambda);
    const float l22 = (float)__builtin_fma(-beta, (double)wz, (double)lambda);
    const float l10 = (float)(beta * (double)sx * (double)sy);
    const float l20 = (float)(beta * (double)sx * (double)sz);
    const float l21 = (float)(beta * (double)sy * (double)sz);
    const float L00 = lg_fma(l20, l20, lg_fma(l00, l00, l10 * l10));
    const float L10 = lg_fma(l20, l21, lg_fma(l00, l10, l10 * l11));
    const float L11 = lg_fma(l21, l21, lg_fma(l10, l10, l11 * l11));
    const float L20 = lg_fma(l20, l22, lg_fma(l00, l20, l10 * l21));
    const float L21 = lg_fma(l21, l22, lg_fma(l10, l20, l11 * l21));
    const float L22 = lg_fma(l22, l22, lg_fma(l20, l20, l21 * l21));
    // split: [kx][r][q < nzc-1] followed by the Nyquist plane [kx][r] (fft_lds.hpp)
    float *t = !split ? tab + (size_t)v.s * 6
               : v.k < nzh ? tab + (((size_t)v.i * g.ny + v.j) * nzh + v.k) * 6
                           : tab + ((size_t)g.nx * g.ny * nzh + (size_t)v.i * g.ny + v.j) * 6;
    if (INV) {  // cuda/metric.cu:47-78
        const float ooG00 = fx_recip(fx_safe_sqrt(L00));
        const float G10 = L10 * ooG00;
        const float G20 = L20 * ooG00;
        float ooG11 = lg_fma(-G10, G10, L11);
        ooG11 = fx_recip(fx_safe_sqrt(ooG11));
        const float G21 = lg_fma(-G20, G10, L21) * ooG11;
        float ooG22 = lg_fma(-G21, G21, lg_fma(-G20, G20, L22));
        ooG22 = fx_recip(fx_safe_sqrt(ooG22));
        t[0] = ooG00; t[1] = G10; t[2] = ooG11; t[3] = G20; t[4] = G21; t[5] = ooG22;
    } else {
        t[0] = L00; t[1] = L10; t[2] = L11; t[3] = L20; t[4] = L21; t[5] = L22;
    }
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

constexpr int kKL = 16;  // lanes along the kz chunk; chunk length KC <= 16
constexpr int kKCP = kKL + 1;  // LDS row pitch in complex elements (odd: spreads rows over banks)

// S consecutive radix-2 levels (halves 2^TOP ... 2^(TOP-S+1)) done in registers on 2^S elements
// spaced 2^(TOP-S+1) apart: one LDS read + one LDS write per element per group instead of per level.
// FWD: decimation in frequency (levels from the largest half down); !FWD: decimation in time with
// conjugated twiddles (levels from the smallest half up).  Same data flow as the radix-2 network,
// so the output order is unchanged (bit-reversed after the forward transform).
template <int LOGN, int TOP, int S, bool FWD>
__device__ __forceinline__ void merged_levels(float2 *buf, const float2 *tw, int row0, int kc) {
    constexpr int NX = 1 << LOGN, R = 1 << S, LH = TOP - S + 1, H = 1 << LH;
    for (int q = row0; q < 3 * NX / R; q += 16) {
        const int c = q / (NX / R), gidx = q % (NX / R);  // powers of two: shifts
        const int low = gidx & (H - 1);
        const int i = ((gidx >> LH) << (LH + S)) | low;
        float2 *p = buf + (c * NX + i) * kKCP + kc;
        float2 v[R];
#pragma unroll
        for (int m = 0; m < R; ++m) v[m] = p[m * H * kKCP];
#pragma unroll
        for (int ll = 0; ll < S; ++ll) {
            const int l = FWD ? ll : S - 1 - ll;
            const int hm = R >> (l + 1);
            const int lhalf = LH + (S - 1 - l);
#pragma unroll
            for (int m = 0; m < R; ++m) {
                if (m & hm) continue;
                const int jj = low + (m & (hm - 1)) * H;
                float2 w = tw[jj << (LOGN - 1 - lhalf)];
                const float2 a = v[m];
                if (FWD) {
                    const float2 bb = v[m + hm];
                    v[m] = make_float2(a.x + bb.x, a.y + bb.y);
                    v[m + hm] = cmul(make_float2(a.x - bb.x, a.y - bb.y), w);
                } else {
                    w.y = -w.y;
                    const float2 bb = cmul(v[m + hm], w);
                    v[m] = make_float2(a.x + bb.x, a.y + bb.y);
                    v[m + hm] = make_float2(a.x - bb.x, a.y - bb.y);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < R; ++m) p[m * H * kKCP] = v[m];
    }
}

// All levels of an NX-point transform in groups of up to three.
template <int LOGN, int TOP, bool FWD>
struct Levels {
    static constexpr int S = TOP + 1 >= 3 ? 3 : TOP + 1;
    static __device__ __forceinline__ void run(float2 *buf, const float2 *tw, int row0, int kc) {
        if (FWD) {
            merged_levels<LOGN, TOP, S, true>(buf, tw, row0, kc);
            __syncthreads();
            Levels<LOGN, TOP - S, FWD>::run(buf, tw, row0, kc);
        } else {
            Levels<LOGN, TOP - S, FWD>::run(buf, tw, row0, kc);
            merged_levels<LOGN, TOP, S, false>(buf, tw, row0, kc);
            __syncthreads();
        }
    }
};
template <int LOGN, bool FWD>
struct Levels<LOGN, -1, FWD> {
    static __device__ __forceinline__ void run(float2 *, const float2 *, int, int) {}
};

// F: (nn, 3, NX, ny, nzc) complex, in place.  Workgroup = (n, y, kz chunk); thread = (row, kc).
template <int LOGN, bool INV>
__global__ __launch_bounds__(256) void fluid_xpass_kernel(float2 *__restrict__ F, const float *__restrict__ tab,
                                                          int ny, int nzc, int KC, int nchunks, float scale,
                                                          uint32_t total, int dbg) {
    constexpr int NX = 1 << LOGN;
    constexpr int KCP = kKCP;
    extern __shared__ __align__(16) unsigned char lago_smem[];
    float2 *buf = reinterpret_cast<float2 *>(lago_smem);          // [3][NX][KCP]
    float2 *tw = buf + 3 * NX * KCP;                               // [NX/2]: exp(-2 pi i t / NX)

    const uint32_t Lb = xcd_swizzle(blockIdx.x, total);
    const uint32_t chunk = Lb % (uint32_t)nchunks;
    const uint32_t ry = Lb / (uint32_t)nchunks;
    const uint32_t y = ry % (uint32_t)ny;
    const uint32_t n = ry / (uint32_t)ny;
    const int k0 = (int)chunk * KC;
    const int kcn = min(KC, nzc - k0);
    const int kc = threadIdx.x & (kKL - 1);
    const int row0 = threadIdx.x >> 4;  // 0..15
    const bool act = kc < kcn;
    const size_t xs = (size_t)ny * nzc;  // stride of x in complex elements
    float2 *Fn = F + (size_t)n * 3 * NX * xs + (size_t)y * nzc + k0 + kc;

    for (int t = threadIdx.x; t < NX / 2; t += 256) {
        double sn, cs;   // (in double, rounded once: fft_lds.hpp twiddle())
        sincospi(-2.0 * (double)t / (double)NX, &sn, &cs);
        tw[t] = make_float2((float)cs, (float)sn);
    }
    // load the bundle: rows r = c*NX + x, kcn contiguous complex each (inactive lanes hold zeros)
    for (int r = row0; r < 3 * NX; r += 16) buf[r * KCP + kc] = act ? Fn[(size_t)r * xs] : make_float2(0.f, 0.f);
    __syncthreads();

    if (dbg != 1) Levels<LOGN, LOGN - 1, true>::run(buf, tw, row0, kc);  // forward: natural -> bit-reversed

    // per-frequency operator; position p holds kx = bitrev(p)
    for (int p = row0; p < NX; p += 16) {
        if (act && dbg != 1 && dbg != 2) {
            const uint32_t kx = __brev((uint32_t)p) >> (32 - LOGN);
            const float *t = dbg == 3 ? tab + kc * 6 : tab + (((size_t)kx * ny + y) * nzc + k0 + kc) * 6;
            const float c0 = t[0], c1 = t[1], c2 = t[2], c3 = t[3], c4 = t[4], c5 = t[5];
            float2 X = buf[(0 * NX + p) * KCP + kc], Y = buf[(1 * NX + p) * KCP + kc], Z = buf[(2 * NX + p) * KCP + kc];
            float bx[2] = {X.x, X.y}, by[2] = {Y.x, Y.y}, bz[2] = {Z.x, Z.y};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float bX = bx[q], bY = by[q], bZ = bz[q];
                if (INV) {  // cuda/metric.cu:103-130 with (ooG00 G10 ooG11 G20 G21 ooG22) = c0..c5
                    const float y0 = bX * c0;
                    const float y1 = lg_fma(-c1, y0, bY) * c2;
                    const float y2 = lg_fma(-c4, y1, lg_fma(-c3, y0, bZ)) * c5;
                    bZ = y2 * c5;
                    bY = lg_fma(-c4, bZ, y1) * c2;
                    bX = lg_fma(-c3, bZ, lg_fma(-c1, bY, y0)) * c0;
                } else {  // cuda/metric.cu:145-160 with (L00 L10 L11 L20 L21 L22) = c0..c5
                    const float x = lg_fma(c3, bZ, lg_fma(c0, bX, c1 * bY));
                    const float yy = lg_fma(c4, bZ, lg_fma(c1, bX, c2 * bY));
                    bZ = lg_fma(c5, bZ, lg_fma(c3, bX, c4 * bY));
                    bX = x;
                    bY = yy;
                }
                bx[q] = bX * scale; by[q] = bY * scale; bz[q] = bZ * scale;
            }
            buf[(0 * NX + p) * KCP + kc] = make_float2(bx[0], bx[1]);
            buf[(1 * NX + p) * KCP + kc] = make_float2(by[0], by[1]);
            buf[(2 * NX + p) * KCP + kc] = make_float2(bz[0], bz[1]);
        }
    }
    __syncthreads();

    if (dbg != 1) Levels<LOGN, LOGN - 1, false>::run(buf, tw, row0, kc);  // inverse: bit-reversed -> natural, unnormalised

    for (int r = row0; r < 3 * NX; r += 16)
        if (act) Fn[(size_t)r * xs] = buf[r * KCP + kc];
}

// ---- host side (called from fft.hip) ---------------------------------------------------------

int fluid_coef_launch(float *tab, int inverse, const float *cosX, const float *sinX, const float *cosY,
                      const float *sinY, const float *cosZ, const float *sinZ, double alpha, double beta,
                      double gamma, int64_t nx, int64_t ny, int64_t nzc, int split, hipStream_t s) {
    Geom g;
    if (!make_geom(g, 3, 1, nx, ny, nzc)) return fail_invalid("fluid_coef: bad extent");
    if (inverse)
        hipLaunchKernelGGL((fluid_coef_kernel<true>), dim3(g.nblocks), dim3(kBlock), 0, s, tab, cosX, sinX, cosY, sinY,
                           cosZ, sinZ, alpha, beta, gamma, g, split);
    else
        hipLaunchKernelGGL((fluid_coef_kernel<false>), dim3(g.nblocks), dim3(kBlock), 0, s, tab, cosX, sinX, cosY, sinY,
                           cosZ, sinZ, alpha, beta, gamma, g, split);
    return finish_launch(s, "fluid_coef");
}

// Profiling builds only (python -m lagomorph_amd.build --profiling -> liblagomorph_hip_prof.so, used by tools/):
// a non-zero variant skips stages of the x pass, i.e. the results are wrong.  The product library has no such knob.
#ifdef LAGO_PROFILING
std::atomic<int> g_xpass_dbg{0};
#define LAGO_XPASS_DBG ((int)g_xpass_dbg)
#else
#define LAGO_XPASS_DBG 0
#endif
bool fluid_xpass_supported(int64_t nx) { return nx == 64 || nx == 128 || nx == 256; }

template <int LOGN>
static hipError_t xpass_launch(float2 *F, const float *tab, int inverse, int64_t nn, int64_t ny, int64_t nzc,
                               float scale, hipStream_t s) {
    constexpr int NX = 1 << LOGN;
    const int nchunks = (int)((nzc + kKL - 1) / kKL);
    const int KC = (int)((nzc + nchunks - 1) / nchunks);
    const size_t smem = (size_t)(3 * NX * (kKL + 1) + NX / 2) * sizeof(float2);
    const uint64_t total = (uint64_t)nn * ny * nchunks;
    if (total >= (1ull << 31)) return hipErrorInvalidValue;
    auto kinv = fluid_xpass_kernel<LOGN, true>;
    auto kfwd = fluid_xpass_kernel<LOGN, false>;
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(inverse ? kinv : kfwd),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
    }
    if (inverse)
        hipLaunchKernelGGL(kinv, dim3((uint32_t)total), dim3(256), smem, s, F, tab, (int)ny, (int)nzc, KC, nchunks, scale,
                           (uint32_t)total, LAGO_XPASS_DBG);
    else
        hipLaunchKernelGGL(kfwd, dim3((uint32_t)total), dim3(256), smem, s, F, tab, (int)ny, (int)nzc, KC, nchunks, scale,
                           (uint32_t)total, LAGO_XPASS_DBG);
    return hipSuccess;
}

int fluid_xpass_launch(float *F, const float *tab, int inverse, int64_t nn, int64_t nx, int64_t ny, int64_t nzc,
                       double scale, hipStream_t s) {
    hipError_t e;
    float2 *Fc = reinterpret_cast<float2 *>(F);
    if (nx == 64) e = xpass_launch<6>(Fc, tab, inverse, nn, ny, nzc, (float)scale, s);
    else if (nx == 128) e = xpass_launch<7>(Fc, tab, inverse, nn, ny, nzc, (float)scale, s);
    else if (nx == 256) e = xpass_launch<8>(Fc, tab, inverse, nn, ny, nzc, (float)scale, s);
    else return fail_invalid("fluid_xpass: unsupported nx");
    if (e != hipSuccess) return fail_hip(e, "fluid_xpass");
    return finish_launch(s, "fluid_xpass");
}

}  // namespace lago

#ifdef LAGO_PROFILING
extern "C" void lago_debug_xpass_variant(int v) { lago::g_xpass_dbg = v; }
#endif

