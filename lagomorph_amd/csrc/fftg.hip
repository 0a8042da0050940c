// Generic hand-written FFT passes for the fluid metric: ANY extent, float32 and float64, 2D and 3D.
//
// The tuned passes of fft3.hip cover float32 volumes whose extents are 2^a, 3*2^a or 5*2^a (and a list of 2D planes);
// every other caller of FluidMetricOperator.forward (lagomorph/metric.py:11-19: it serves any shape and dtype) used to
// fall back to rocFFT -- a third-party library that ROCm 7.2 ships with a batched 2D real transform that can come back
// 60 % wrong (tools/probes/rocfft_2d_repro.py).  This file is the fallback that needs no library: float64 (the
// reference's gradcheck suite), extents with other prime factors (120, 182 x 218 x 182, ...), 2D planes beyond the LDS.
// Generality first, then speed: one kernel transforms batches of LINES along one axis --
//   * a workgroup brings L lines of N points into LDS as [point][line] (adjacent lines of a strided axis are adjacent
//     in memory: coalesced rows; L from lines_pass),
//   * runs a Stockham autosort FFT with the radices of N's factorisation (8 and 4 where they divide, then the primes
//     ascending).  Radices 2, 3, 4, 5, 7: one whole butterfly per thread in registers (r LDS reads, r writes, r - 1
//     twiddles); any other prime: a direct r-point DFT, one thread per conjugate pair of outputs -- so ANY extent works; a line whose
//     length has a prime factor of 29 or more goes through Bluestein's identity instead (chirp, two power-of-two
//     transforms of M >= 2 N - 1 points, a cached table): 182 x 218 x 182 and 193 x 229 x 193 volumes 2 - 5x faster,
//   * lengths whose factors are 8, 4 and 2 only (and every Bluestein transform) run an instantiation without the odd
//     radices (half the registers); long strided power-of-two lines run their stages IN PLACE (stage_inplace: the
//     butterflies of a stage held in registers across a barrier, one line buffer instead of two),
//   * one table of the N-th roots per workgroup (sincospi of the exactly reduced argument, in double for float64 lines
//     and in float for float32 ones); stage twiddles and the r-th roots are strided reads of it,
//   * the real axis packs TWO real lines into one complex line (a + i b) and separates / rebuilds the two half spectra
//     by conjugate symmetry; pairs never cross a field, so a batch item's bits do not depend on its neighbours.
// Measured against rocFFT on the shapes it used to serve (profiles/r04_fft_generic.md): 0.52 - 1.55x its time.
// The per-frequency operator between the passes is metric.hip's fluid_kernel, the same as on the rocFFT path, with
// the 1/N of the unnormalised transform pair folded in.  Layout of the half spectrum: [n][c][x][y][z <= nz/2] complex,
// what rocFFT's R2C produces, so the operator kernel does not know which path ran.
#include <map>
#include <memory>
#include <tuple>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "fluid_bin.hpp"

namespace lago {

template <typename R>
int fluid_operator_impl(R *Fm, int inverse, const R *cosX, const R *sinX, const R *cosY, const R *sinY,
                        const R *cosZ, const R *sinZ, double alpha, double beta, double gamma, int dim,
                        int64_t nn, int64_t nx, int64_t ny, int64_t nz, void *stream, double scale);  // metric.hip

template <typename R>
struct alignas(2 * sizeof(R)) GC {
    R re, im;
};
template <typename R>
__device__ __forceinline__ GC<R> cmul(GC<R> a, GC<R> b) {
    return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}

template <typename R>
__device__ __forceinline__ GC<R> cadd(GC<R> a, GC<R> b) { return {a.re + b.re, a.im + b.im}; }
template <typename R>
__device__ __forceinline__ GC<R> csub(GC<R> a, GC<R> b) { return {a.re - b.re, a.im - b.im}; }
// a * (sign i)
template <typename R>
__device__ __forceinline__ GC<R> cmuli(GC<R> a, int sign) {
    return sign > 0 ? GC<R>{-a.im, a.re} : GC<R>{a.im, -a.re};
}

template <typename R>
__device__ __forceinline__ GC<R> root(int k, int n, int sign);   // exp(sign 2 pi i k / n)
template <>
__device__ __forceinline__ GC<double> root<double>(int k, int n, int sign) {
    double sn, cs;
    sincospi(2.0 * (double)k / (double)n, &sn, &cs);
    return {cs, (double)sign * sn};
}
template <>
__device__ __forceinline__ GC<float> root<float>(int k, int n, int sign) {
    double sn, cs;   // in double, rounded once (fft_lds.hpp twiddle(): sincospif of the rounded argument is off by up to six ulps)
    sincospi(2.0 * (double)k / (double)n, &sn, &cs);
    return {(float)cs, (float)sign * (float)sn};
}

struct GLines {
    int N;               // points per line
    uint32_t inner;      // mode 0: stride between consecutive points of a line (elements) = lines adjacent in memory
    uint32_t nlines;     // mode 0: lines of the launch.  modes 1 / 2: lines per plane (one (n, c) field: X * Y)
    uint32_t ppp;        // modes 1 / 2: PAIRS of lines per plane = ceil(nlines / 2), chunks of L pairs per plane
    uint32_t chunks;
    int L, Lp;           // lines (mode 0) or pairs (modes 1 / 2) per workgroup; Lp = L | 1, the LDS row pitch
    int sign;            // -1 forward, +1 inverse (unnormalised)
    int mode;            // 0 complex -> complex in place; 1 real lines -> half spectra; 2 half spectra -> real lines
    int nhalf;           // mode 1 / 2: N / 2 + 1
    int nfac;
    int fac[20];
    FastDiv dinner, dN, dL, dnhalf, dchunks;
    FastDiv ds[20], dr[20];   // per stage: division by the stride s and by (r + 1) / 2 (the items of a direct-DFT butterfly)
    // Bluestein (M > 0): the N-point transform as a circular convolution of length M = 2^k >= 2 N - 1; fac / ds then
    // describe the M-point transform (radix 4 and 2 only) and bhat is FFT_M of the conjugate chirp, 1 / M folded in
    int M;
    const void *bhat;
    int inplace;         // power-of-two stages in place (one line buffer; host: L N <= 8192)
};

// line l of a mode-0 launch: element offset of its first point
__device__ __forceinline__ size_t gline_base(const GLines &a, uint32_t l, int n_points) {
    const uint32_t hi = a.dinner.div(l), lo = l - hi * a.inner;
    return (size_t)hi * (size_t)n_points * a.inner + lo;
}

// radix-2 / 4 / 8 butterflies on registers: o[u] = sum_t v[t] w^(t u), w = exp(sgn 2 pi i / RR)
template <typename R, int RR>
__device__ __forceinline__ void bfly_pow2(const GC<R> (&v)[RR], GC<R> (&o)[RR], int sgn) {
    typedef GC<R> C;
    if constexpr (RR == 2) {
        o[0] = cadd(v[0], v[1]);
        o[1] = csub(v[0], v[1]);
    } else if constexpr (RR == 8) {
        // two 4-point transforms of the even and the odd inputs, then X[u] = E[u] + w8^u O[u], X[u + 4] = E[u] - w8^u O[u]
        // with w8 = exp(sgn 2 pi i / 8): w8^1 = (1 + sgn i) / sqrt 2, w8^2 = sgn i, w8^3 = (-1 + sgn i) / sqrt 2
        const C a0 = cadd(v[0], v[4]), a1 = csub(v[0], v[4]), a2 = cadd(v[2], v[6]), a3 = cmuli(csub(v[2], v[6]), sgn);
        const C b0 = cadd(v[1], v[5]), b1 = csub(v[1], v[5]), b2 = cadd(v[3], v[7]), b3 = cmuli(csub(v[3], v[7]), sgn);
        const C E0 = cadd(a0, a2), E1 = cadd(a1, a3), E2 = csub(a0, a2), E3 = csub(a1, a3);
        const C O0 = cadd(b0, b2), O1 = cadd(b1, b3), O2 = csub(b0, b2), O3 = csub(b1, b3);
        const R h = (R)0.70710678118654752440;
        const C i1 = cmuli(O1, sgn), i3 = cmuli(O3, sgn);
        const C t1 = {h * (O1.re + i1.re), h * (O1.im + i1.im)};      // w8^1 O1
        const C t2 = cmuli(O2, sgn);                                  // w8^2 O2
        const C t3 = {h * (i3.re - O3.re), h * (i3.im - O3.im)};      // w8^3 O3
        o[0] = cadd(E0, O0); o[4] = csub(E0, O0);
        o[1] = cadd(E1, t1); o[5] = csub(E1, t1);
        o[2] = cadd(E2, t2); o[6] = csub(E2, t2);
        o[3] = cadd(E3, t3); o[7] = csub(E3, t3);
    } else {
        static_assert(RR == 4, "bfly_pow2: radix 2, 4 or 8");
        const C e0 = cadd(v[0], v[2]), e1 = csub(v[0], v[2]), o0 = cadd(v[1], v[3]), o1 = cmuli(csub(v[1], v[3]), sgn);
        o[0] = cadd(e0, o0);
        o[1] = cadd(e1, o1);
        o[2] = csub(e0, o0);
        o[3] = csub(e1, o1);
    }
}

// One Stockham stage of radix RR on every line of the workgroup, a whole butterfly per thread in registers:
//   y[q + s (RR p + u)] = (sum_t x[q + s (p + t m)] w_RR^(t u)) w_n^(p u),  n = RR m the remaining length, s the stride;
// w_n^(p u) = W[p u s] and w_RR^k = W[k N / RR] from the one table of N-th roots.  LDS layout [point][line]: the line
// index runs fastest over the threads, so every read and write is to consecutive addresses and the twiddle is a
// broadcast.
// `pre` (Bluestein's second transform, first stage): input point i is taken as conj(x[i] pre[i]).
template <typename R, int RR>
__device__ __forceinline__ void stage_fixed(const GC<R> *__restrict__ x, GC<R> *__restrict__ y, const GC<R> *__restrict__ W,
                                            const GLines &a, int N, int sgn, int nl, int s, int m, FastDiv ds,
                                            const GC<R> *__restrict__ pre = nullptr) {
    typedef GC<R> C;
    const int Lp = a.Lp, L = a.L;
    C wr[RR];
    if (RR != 2 && RR != 4 && RR != 8) {
#pragma unroll
        for (int t = 0; t < RR; ++t) wr[t] = W[t * (N / RR)];
    }
    const int nb = (N / RR) * L;
    for (int b = threadIdx.x; b < nb; b += kBlock) {
        const int j = (int)a.dL.div((uint32_t)b), ln = b - j * L;
        if (ln >= nl) continue;
        const int p = (int)ds.div((uint32_t)j), q = j - p * s;
        const C *xi = x + (size_t)(q + s * p) * Lp + ln;
        C v[RR], o[RR];
#pragma unroll
        for (int t = 0; t < RR; ++t) v[t] = xi[(size_t)s * m * t * Lp];
        if (pre) {   // (workgroup-uniform)
#pragma unroll
            for (int t = 0; t < RR; ++t) {
                const C pr = cmul(v[t], pre[q + s * (p + t * m)]);
                v[t] = {pr.re, -pr.im};
            }
        }
        if constexpr (RR == 2 || RR == 4 || RR == 8) {
            bfly_pow2<R, RR>(v, o, sgn);
        } else {
            // odd radix: X[0] the plain sum; the outputs u and RR - u share their products (conjugate roots, see stage_any)
            C s0 = v[0];
#pragma unroll
            for (int t = 1; t < RR; ++t) s0 = cadd(s0, v[t]);
            o[0] = s0;
#pragma unroll
            for (int u = 1; u <= (RR - 1) / 2; ++u) {
                R P = v[0].re, Q = 0, Rr = 0, T = v[0].im;
#pragma unroll
                for (int t = 1; t < RR; ++t) {
                    const C w = wr[(t * u) % RR];
                    P = __builtin_fma(v[t].re, w.re, P);
                    Q = __builtin_fma(v[t].im, w.im, Q);
                    Rr = __builtin_fma(v[t].re, w.im, Rr);
                    T = __builtin_fma(v[t].im, w.re, T);
                }
                o[u] = {P - Q, Rr + T};
                o[RR - u] = {P + Q, T - Rr};
            }
        }
        C *yo = y + (size_t)(q + s * RR * p) * Lp + ln;
        yo[0] = o[0];
#pragma unroll
        for (int u = 1; u < RR; ++u) yo[(size_t)s * u * Lp] = cmul(o[u], W[p * u * s]);
    }
}

// any other radix (the odd primes from 11 up): a direct r-point DFT, one thread per PAIR of outputs (u, r - u) -- their
// twiddles are complex conjugates, so the pair shares its r inputs and r roots and costs 4 instead of 8 real
// multiply-adds per input: with x = a + i b and w = c + i s,  X[u] = (P - Q) + i (R + T),  X[r - u] = (P + Q) + i (T - R),
// P = sum a c, Q = sum b s, R = sum a s, T = sum b c.  u = 0 (the plain sum) is an item of its own.
template <typename R>
__device__ __forceinline__ void stage_any(const GC<R> *__restrict__ x, GC<R> *__restrict__ y, const GC<R> *__restrict__ W,
                                          const GLines &a, int N, int nl, int r, int s, int m, FastDiv ds, FastDiv dr) {
    typedef GC<R> C;
    const int Lp = a.Lp, L = a.L, wstep = N / r, h = (r + 1) / 2;   // h items per butterfly: u = 0 and (r - 1) / 2 pairs
    const FastDiv dh = dr;
    const int items = m * s * h * L;   // butterflies (m s) x items x lines
    for (int i = threadIdx.x; i < items; i += kBlock) {
        const int w0 = (int)a.dL.div((uint32_t)i), ln = i - w0 * L;
        if (ln >= nl) continue;
        const int j = (int)dh.div((uint32_t)w0), u = w0 - j * h;   // butterfly j = q + s p, item u
        const int p = (int)ds.div((uint32_t)j), q = j - p * s;
        const C *xi = x + (size_t)(q + s * p) * Lp + ln;
        C *yo = y + (size_t)(q + s * r * p) * Lp + ln;
        if (u == 0) {
            C acc = {(R)0, (R)0};
            for (int t = 0; t < r; ++t) {
                const C v = xi[(size_t)s * m * t * Lp];
                acc.re += v.re;
                acc.im += v.im;
            }
            yo[0] = acc;
            continue;
        }
        R P = 0, Q = 0, Rr = 0, T = 0;
        int tu = 0;   // (t u) mod r
        for (int t = 0; t < r; ++t) {
            const C v = xi[(size_t)s * m * t * Lp], w = W[tu * wstep];
            P = __builtin_fma(v.re, w.re, P);
            Q = __builtin_fma(v.im, w.im, Q);
            Rr = __builtin_fma(v.re, w.im, Rr);
            T = __builtin_fma(v.im, w.re, T);
            tu += u;
            if (tu >= r) tu -= r;
        }
        yo[(size_t)s * u * Lp] = cmul(C{P - Q, Rr + T}, W[p * u * s]);
        yo[(size_t)s * (r - u) * Lp] = cmul(C{P + Q, T - Rr}, W[p * (r - u) * s]);
    }
}

// The same stage IN PLACE for the power-of-two radices: every thread first reads all its butterflies (at most 32 complex
// numbers: 32 / RR butterflies) into registers, a barrier, then writes them to their Stockham positions in the SAME
// buffer -- half the LDS of the ping-pong form, which is what long strided lines need (1024 points x 8 lines of
// float32: 64 KB instead of 128: two workgroups per CU, or twice the adjacent lines).  The host keeps
// (N / RR) L <= 256 * (32 / RR), i.e. L N <= 8192.
template <typename R, int RR>
__device__ __forceinline__ void stage_inplace(GC<R> *__restrict__ x, const GC<R> *__restrict__ W, const GLines &a, int N, int sgn,
                                              int nl, int s, int m, FastDiv ds, const GC<R> *__restrict__ pre) {
    typedef GC<R> C;
    constexpr int TM = 32 / RR;
    const int Lp = a.Lp, L = a.L, nb = (N / RR) * L;
    C v[TM][RR];
#pragma unroll
    for (int tr = 0; tr < TM; ++tr) {
        const int b = (int)threadIdx.x + tr * kBlock;
        if (b < nb) {
            const int j = (int)a.dL.div((uint32_t)b), ln = b - j * L;
            const int p = (int)ds.div((uint32_t)j), q = j - p * s;
            const C *xi = x + (size_t)(q + s * p) * Lp + ln;
#pragma unroll
            for (int t = 0; t < RR; ++t) v[tr][t] = xi[(size_t)s * m * t * Lp];
            if (pre) {
#pragma unroll
                for (int t = 0; t < RR; ++t) {
                    const C pr = cmul(v[tr][t], pre[q + s * (p + t * m)]);
                    v[tr][t] = {pr.re, -pr.im};
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int tr = 0; tr < TM; ++tr) {
        const int b = (int)threadIdx.x + tr * kBlock;
        if (b < nb) {
            const int j = (int)a.dL.div((uint32_t)b), ln = b - j * L;
            if (ln < nl) {
                const int p = (int)ds.div((uint32_t)j), q = j - p * s;
                C o[RR];
                bfly_pow2<R, RR>(v[tr], o, sgn);
                C *yo = x + (size_t)(q + s * RR * p) * Lp + ln;
                yo[0] = o[0];
#pragma unroll
                for (int u = 1; u < RR; ++u) yo[(size_t)s * u * Lp] = cmul(o[u], W[p * u * s]);
            }
        }
    }
}

// All Stockham stages of an NP-point transform (factors a.fac, roots W of sign sgn) on the workgroup's lines; returns
// the buffer that holds the result (x and y alternate).  Begins with a barrier (the caller's writes to x and W) and
// ends with one.
// RMAX = 4: the factors are 4 and 2 only (power-of-two lengths, every Bluestein line) -- the kernel then carries neither
// the odd-radix butterflies nor the direct stage and needs about half the registers: more workgroups per CU.
template <typename R, int RMAX, bool INPL = false>
__device__ __forceinline__ GC<R> *run_stages(GC<R> *x, GC<R> *y, const GC<R> *W, const GLines &a, int NP, int sgn, int nl,
                                             const GC<R> *pre = nullptr) {
    int n = NP, s = 1;
    for (int f = 0; f < a.nfac; ++f) {
        const int r = a.fac[f], m = n / r;
        __syncthreads();
        if constexpr (RMAX <= 4) {   // (power-of-two lengths: radix 8, 4, 2)
            const GC<R> *pf = f == 0 ? pre : nullptr;
            if constexpr (INPL) {   // (y == x)
                if (r == 8) stage_inplace<R, 8>(x, W, a, NP, sgn, nl, s, m, a.ds[f], pf);
                else if (r == 4) stage_inplace<R, 4>(x, W, a, NP, sgn, nl, s, m, a.ds[f], pf);
                else stage_inplace<R, 2>(x, W, a, NP, sgn, nl, s, m, a.ds[f], pf);
            } else if (r == 8) stage_fixed<R, 8>(x, y, W, a, NP, sgn, nl, s, m, a.ds[f], pf);
            else if (r == 4) stage_fixed<R, 4>(x, y, W, a, NP, sgn, nl, s, m, a.ds[f], pf);
            else stage_fixed<R, 2>(x, y, W, a, NP, sgn, nl, s, m, a.ds[f], pf);
        } else {
            switch (r) {
                case 2: stage_fixed<R, 2>(x, y, W, a, NP, sgn, nl, s, m, a.ds[f]); break;
                case 3: stage_fixed<R, 3>(x, y, W, a, NP, sgn, nl, s, m, a.ds[f]); break;
                case 4: stage_fixed<R, 4>(x, y, W, a, NP, sgn, nl, s, m, a.ds[f]); break;
                case 5: stage_fixed<R, 5>(x, y, W, a, NP, sgn, nl, s, m, a.ds[f]); break;
                case 7: stage_fixed<R, 7>(x, y, W, a, NP, sgn, nl, s, m, a.ds[f]); break;
                case 8: stage_fixed<R, 8>(x, y, W, a, NP, sgn, nl, s, m, a.ds[f]); break;
                case 11:
                    if constexpr (RMAX >= 13) { stage_fixed<R, 11>(x, y, W, a, NP, sgn, nl, s, m, a.ds[f]); break; }
                    [[fallthrough]];
                case 13:
                    if constexpr (RMAX >= 13) {
                        if (r == 13) { stage_fixed<R, 13>(x, y, W, a, NP, sgn, nl, s, m, a.ds[f]); break; }
                    }
                    [[fallthrough]];
                default: stage_any<R>(x, y, W, a, NP, nl, r, s, m, a.ds[f], a.dr[f]); break;
            }
        }
        GC<R> *tmp = x; x = y; y = tmp;
        n = m;
        s *= r;
    }
    __syncthreads();
    return x;
}

// exp(sign pi i n^2 / N): the chirp of Bluestein's identity n k = (n^2 + k^2 - (k - n)^2) / 2, the argument reduced
// exactly (n^2 mod 2 N in integers)
template <typename R>
__device__ __forceinline__ GC<R> chirp(int n, int N, int sign) {
    const unsigned long long q = ((unsigned long long)n * (unsigned long long)n) % (2ull * (unsigned long long)N);
    if constexpr (sizeof(R) == 8) {
        double sn, cs;
        sincospi((double)q / (double)N, &sn, &cs);
        return {cs, (double)sign * sn};
    } else {
        double sn, cs;
        sincospi((double)q / (double)N, &sn, &cs);
        return {(float)cs, (float)sign * (float)sn};
    }
}

// bhat[k] = FFT_M(d)[k] / M with d[m] = conj(chirp)(|m|) for |m| < N (indices mod M), 0 elsewhere: one workgroup, once
// per (N, sign, precision) -- lines_pass caches the table
template <typename R>
__global__ __launch_bounds__(kBlock) void bluestein_table_kernel(GC<R> *__restrict__ bhat, GLines a) {
    extern __shared__ __align__(16) unsigned char lago_fg[];
    typedef GC<R> C;
    const int N = a.N, M = a.M;
    C *x = reinterpret_cast<C *>(lago_fg), *y = x + M, *W = y + M;   // (a.L = a.Lp = 1)
    for (int k = threadIdx.x; k < M; k += kBlock) {
        W[k] = root<R>(k, M, -1);
        const int idx = k < N ? k : (M - k < N ? M - k : -1);
        C d = {(R)0, (R)0};
        if (idx >= 0) { d = chirp<R>(idx, N, a.sign); d.im = -d.im; }
        x[k] = d;
    }
    const C *res = run_stages<R, 4>(x, y, W, a, M, -1, 1);
    const R inv = (R)(1.0 / (double)M);
    for (int k = threadIdx.x; k < M; k += kBlock) bhat[k] = {res[k].re * inv, res[k].im * inv};
}

template <typename R, int RMAX, bool INPL = false>
__global__ __launch_bounds__(kBlock) void fft_lines_kernel(GC<R> *__restrict__ spec, const R *__restrict__ rin, R *__restrict__ rout,
                                                           GLines a) {
    extern __shared__ __align__(16) unsigned char lago_fg[];
    typedef GC<R> C;
    const int N = a.N, L = a.L, Lp = a.Lp;
    const int NB = a.M ? a.M : N;   // rows of the two line buffers and entries of the root table
    C *x = reinterpret_cast<C *>(lago_fg), *y = INPL ? x : x + (size_t)Lp * NB, *W = y + (size_t)Lp * NB;
    // modes 1 / 2: the workgroup's pairs [j0, j0 + nl) of plane `plane`; lines 2 j and 2 j + 1 share one complex line
    uint32_t l0 = 0, plane = 0;
    int nl;
    if (a.mode == 0) {
        l0 = blockIdx.x * (uint32_t)L;
        nl = (int)min((uint32_t)L, a.nlines - l0);
    } else {
        plane = a.dchunks.div(blockIdx.x);
        l0 = (blockIdx.x - plane * a.chunks) * (uint32_t)L;
        nl = (int)min((uint32_t)L, a.ppp - l0);
    }
    const size_t pline = (size_t)plane * a.nlines;   // first line of the plane
    // Bluestein: the chirp first -- the lines are multiplied by it while they are loaded, rows N .. M - 1 are zero
    C *Cq = W + NB;   // (N entries behind the root table)
    if (a.M) {
        for (int k = threadIdx.x; k < N; k += kBlock) Cq[k] = chirp<R>(k, N, a.sign);
        for (int i = threadIdx.x; i < L * (a.M - N); i += kBlock) {
            const int pt = (int)a.dL.div((uint32_t)i), ln = i - pt * L;
            x[(size_t)(N + pt) * Lp + ln] = C{(R)0, (R)0};
        }
        __syncthreads();
    }
    auto put = [&](int pt, int ln, C v) { x[(size_t)pt * Lp + ln] = a.M ? cmul(v, Cq[pt]) : v; };
    // ---- load into x[point][line]
    if (a.mode == 1) {            // two real lines -> one complex line (a + i b)
        for (int i = threadIdx.x; i < nl * N; i += kBlock) {
            const int pr = (int)a.dN.div((uint32_t)i), pt = i - pr * N;
            const uint32_t la = 2 * (l0 + pr);
            const R *ra = rin + (pline + la) * N;
            put(pt, pr, C{ra[pt], la + 1 < a.nlines ? ra[N + pt] : (R)0});
        }
    } else if (a.mode == 2) {     // two half spectra A, B -> the full spectrum of a + i b: Z[k] = A[k] + i B[k],
                                  // A[N - k] = conj A[k]; the imaginary parts of the self-conjugate bins are ignored,
                                  // as a complex-to-real transform does
        for (int i = threadIdx.x; i < nl * N; i += kBlock) {
            const int pr = (int)a.dN.div((uint32_t)i), pt = i - pr * N;
            const uint32_t la = 2 * (l0 + pr);
            const bool up = pt >= a.nhalf;
            const int k = up ? N - pt : pt;
            const C *sa = spec + (pline + la) * a.nhalf;
            C A = sa[k], B = la + 1 < a.nlines ? sa[a.nhalf + k] : C{(R)0, (R)0};
            if (k == 0 || 2 * k == N) A.im = B.im = (R)0;
            if (up) { A.im = -A.im; B.im = -B.im; }
            put(pt, pr, C{A.re - B.im, A.im + B.re});
        }
    } else {                      // strided complex lines: adjacent lines are adjacent in memory
        for (int i = threadIdx.x; i < L * N; i += kBlock) {
            const int pt = (int)a.dL.div((uint32_t)i), ln = i - pt * L;
            if (ln < nl) put(pt, ln, spec[gline_base(a, l0 + ln, N) + (size_t)pt * a.inner]);
        }
    }
    if (a.M == 0) {
        // the N-th roots of unity, once (argument reduced exactly, k / N with k < N; sincospi in double, rounded
        // once to the line's precision)
        for (int k = threadIdx.x; k < N; k += kBlock) W[k] = root<R>(k, N, a.sign);
        x = run_stages<R, RMAX, INPL>(x, y, W, a, N, a.sign, nl);
    } else {
        // Bluestein: X[k] = c[k] sum_n (x[n] c[n]) conj(c)[k - n], c = chirp: a circular convolution of length M through
        // two M-point power-of-two transforms; the inverse one as conj(FFT(conj(.))) with the same roots.  The product
        // with the table is taken by the second transform's first stage as it reads, the final conj(.) c[k] by the store.
        const int M = a.M;
        for (int k = threadIdx.x; k < M; k += kBlock) W[k] = root<R>(k, M, -1);
        C *res = run_stages<R, 4, INPL>(x, y, W, a, M, -1, nl);
        C *other = res == x ? y : x;
        x = run_stages<R, 4, INPL>(res, other, W, a, M, -1, nl, reinterpret_cast<const C *>(a.bhat));
    }
    auto get = [&](int pt, int ln) {
        const C v = x[(size_t)pt * Lp + ln];
        return a.M ? cmul(C{v.re, -v.im}, Cq[pt]) : v;
    };
    // ---- store
    if (a.mode == 1) {            // A[k] = (Z[k] + conj Z[N - k]) / 2,  B[k] = (Z[k] - conj Z[N - k]) / (2 i)
        for (int i = threadIdx.x; i < nl * a.nhalf; i += kBlock) {
            const int pr = (int)a.dnhalf.div((uint32_t)i), k = i - pr * a.nhalf;
            const uint32_t la = 2 * (l0 + pr);
            const C zk = get(k, pr), zm = get(k ? N - k : 0, pr);
            C *sa = spec + (pline + la) * a.nhalf;
            sa[k] = {(R)0.5 * (zk.re + zm.re), (R)0.5 * (zk.im - zm.im)};
            if (la + 1 < a.nlines) sa[a.nhalf + k] = {(R)0.5 * (zk.im + zm.im), (R)0.5 * (zm.re - zk.re)};
        }
    } else if (a.mode == 2) {
        for (int i = threadIdx.x; i < nl * N; i += kBlock) {
            const int pr = (int)a.dN.div((uint32_t)i), pt = i - pr * N;
            const uint32_t la = 2 * (l0 + pr);
            const C z = get(pt, pr);
            R *ra = rout + (pline + la) * N;
            ra[pt] = z.re;
            if (la + 1 < a.nlines) ra[N + pt] = z.im;
        }
    } else {
        for (int i = threadIdx.x; i < L * N; i += kBlock) {
            const int pt = (int)a.dL.div((uint32_t)i), ln = i - pt * L;
            if (ln < nl) spec[gline_base(a, l0 + ln, N) + (size_t)pt * a.inner] = get(pt, ln);
        }
    }
}

// The x pass of the 3D operator with the operator inside (round 6): forward transform along x, the per-frequency 3 x 3
// operator on the three components, inverse transform -- ONE read and ONE write of the half spectrum instead of three of
// each (x forward, operator kernel, x inverse): 5 instead of 7 passes per fluid_metric call on the generic path.  A
// workgroup takes L adjacent (y, z-bin) columns of one batch item, all three components: 3 L lines in LDS, line index
// 3 j + c (a column's components adjacent, so the valid lines of a ragged last chunk are contiguous).  The stages are
// those of fft_lines_kernel (run_stages on the same roots), the operator is fluid_bin.hpp's: the bits of the three
// separate launches.  Lengths that run as Bluestein convolutions keep the separate passes (host); power-of-two lines of 256
// points and more run their stages in place (INPL: one line buffer, twice the adjacent columns in the same LDS).
struct XopArgs {
    const void *cosX, *sinX, *cosY, *sinY, *cosZ, *sinZ;
    double alpha, beta, gamma, scale;
    uint32_t cols;      // columns per component plane: Y * zc
    uint32_t zc;        // bins of the last axis
    uint32_t chunks;    // chunks of L columns per batch item
    FastDiv dchunks, dzc, dLc;
    int Lc;             // columns per workgroup (a.L = 3 Lc lines)
};

template <typename R, int RMAX, bool INV, int DIM, bool INPL = false>
__global__ __launch_bounds__(kBlock) void fft_xop_kernel(GC<R> *__restrict__ spec, GLines a, XopArgs o) {
    extern __shared__ __align__(16) unsigned char lago_fg[];
    typedef GC<R> C;
    const int N = a.N, Lp = a.Lp, Lc = o.Lc;
    C *x = reinterpret_cast<C *>(lago_fg), *y = INPL ? x : x + (size_t)Lp * N, *W = y + (size_t)Lp * N;
    const uint32_t n = o.dchunks.div(blockIdx.x);
    const uint32_t c0 = (blockIdx.x - n * o.chunks) * (uint32_t)Lc;      // first column
    const int ncol = (int)min((uint32_t)Lc, o.cols - c0);
    const int nl = DIM * ncol;
    const size_t planeC = (size_t)N * o.cols;                            // complex elements per component
    C *base = spec + (size_t)n * DIM * planeC + c0;
    // load: (component, point, column) with the column fastest over the threads
    for (int i = threadIdx.x; i < DIM * N * Lc; i += kBlock) {
        const int r = (int)o.dLc.div((uint32_t)i), j = i - r * Lc;
        const int c = (int)a.dN.div((uint32_t)r), pt = r - c * N;
        if (j < ncol) x[(size_t)pt * Lp + DIM * j + c] = base[(size_t)c * planeC + (size_t)pt * o.cols + j];
    }
    for (int k = threadIdx.x; k < N; k += kBlock) W[k] = root<R>(k, N, -1);
    C *res = run_stages<R, RMAX, INPL>(x, y, W, a, N, -1, nl);
    // operator: the frequency along the transformed axis = point index (the stages return natural order), the others from
    // the column.  3D: (kx, ky, kz) = (point, column / zc, column % zc); 2D (geometry (1, nx, ny): the transformed axis is
    // the fields' first one, LUT "X"): (point, column)
    const R *cX = (const R *)o.cosX, *sX = (const R *)o.sinX, *cY = (const R *)o.cosY, *sY = (const R *)o.sinY,
            *cZ = (const R *)o.cosZ, *sZ = (const R *)o.sinZ;
    const R scale = (R)o.scale;
    for (int i = threadIdx.x; i < N * Lc; i += kBlock) {
        const int kx = (int)o.dLc.div((uint32_t)i), j = i - kx * Lc;
        if (j < ncol) {
            const uint32_t col = c0 + (uint32_t)j;
            C *q = res + (size_t)kx * Lp + DIM * j;
            if constexpr (DIM == 3) {
                const uint32_t ky = o.dzc.div(col), kz = col - ky * o.zc;
                FluidBin3<R, INV> op;
                op.setup(cX[kx], cY[ky], cZ[kz], sX[kx], sY[ky], sZ[kz], o.alpha, o.beta, o.gamma);
                C A = q[0], B = q[1], Cc = q[2];
                op.apply(A.re, B.re, Cc.re);
                op.apply(A.im, B.im, Cc.im);
                q[0] = C{A.re * scale, A.im * scale};   // scale == 1 is a bitwise no-op (as in the operator kernel)
                q[1] = C{B.re * scale, B.im * scale};
                q[2] = C{Cc.re * scale, Cc.im * scale};
            } else {
                FluidBin2<R, INV> op;
                op.setup(cX[kx], cY[col], sX[kx], sY[col], o.alpha, o.beta, o.gamma);
                C A = q[0], B = q[1];
                op.apply(A.re, B.re);
                op.apply(A.im, B.im);
                q[0] = C{A.re * scale, A.im * scale};
                q[1] = C{B.re * scale, B.im * scale};
            }
        }
    }
    // the inverse transform on the conjugate roots (run_stages begins with a barrier: the operator's writes and these)
    __syncthreads();
    for (int k = threadIdx.x; k < N; k += kBlock) W[k] = root<R>(k, N, +1);
    C *other = res == x ? y : x;
    res = run_stages<R, RMAX, INPL>(res, other, W, a, N, +1, nl);
    for (int i = threadIdx.x; i < DIM * N * Lc; i += kBlock) {
        const int r = (int)o.dLc.div((uint32_t)i), j = i - r * Lc;
        const int c = (int)a.dN.div((uint32_t)r), pt = r - c * N;
        if (j < ncol) base[(size_t)c * planeC + (size_t)pt * o.cols + j] = res[(size_t)pt * Lp + DIM * j + c];
    }
}

// power-of-two length: radix-8 stages first (512 = 8 * 8 * 8: three LDS round trips instead of five), then 4, then 2
static void factorise_pow2(int N, GLines &a) {
    a.nfac = 0;
    int n = N;
    while (n % 8 == 0) { a.fac[a.nfac++] = 8; n /= 8; }
    while (n % 4 == 0) { a.fac[a.nfac++] = 4; n /= 4; }
    while (n % 2 == 0) { a.fac[a.nfac++] = 2; n /= 2; }
}

static void factorise(int N, GLines &a) {
    if (N > 1 && (N & (N - 1)) == 0) { factorise_pow2(N, a); return; }
    a.nfac = 0;
    int n = N;
    while (n % 8 == 0) { a.fac[a.nfac++] = 8; n /= 8; }   // (the power of two in radix-8 stages first: 120 = 8 * 3 * 5)
    while (n % 4 == 0) { a.fac[a.nfac++] = 4; n /= 4; }
    for (int p = 2; (long long)p * p <= n; ++p)
        while (n % p == 0) { a.fac[a.nfac++] = p; n /= p; }
    if (n > 1) a.fac[a.nfac++] = n;
}

static int largest_prime_factor(int n) {
    int best = 1;
    for (int p = 2; (long long)p * p <= n; ++p)
        while (n % p == 0) { best = p; n /= p; }
    return n > 1 ? n : best;
}
// LDS bytes of ONE Bluestein line in the ping-pong form (lines_pass below: (2 (L | 1) + 1) M + N complex numbers at
// L = 1), M >= 2 N - 1 a power of two.  (lines_pass takes the in-place form only to widen a pass that already fits.)
static size_t bluestein_lds(int N, size_t cb) {
    int M = 1;
    while (M < 2 * N - 1) M <<= 1;
    return ((size_t)3 * M + N) * cb;
}

// three line buffers' worth of complex numbers (ping-pong + roots) must fit the LDS for ONE line: 4096 points of
// float32, 2048 of float64.  A line with a prime factor of 29 or more runs as a Bluestein convolution, whose
// power-of-two length M >= 2 N - 1 needs its own room; where even one such line does not fit (N above 2048 in
// float32, above 1024 in float64) the direct-DFT stage would cost N r multiply-adds per line (r = 4093: 17 M per
// line) -- such a shape is NOT taken here and goes to the guarded rocFFT plan.
bool fluid_generic_supported(int dim, int64_t nx, int64_t ny, int64_t nz, size_t esize) {
    const int64_t maxn = esize == 4 ? 4096 : 2048;
    const int64_t ext[3] = {dim == 3 ? nx : 1, dim == 3 ? ny : nx, dim == 3 ? nz : ny};
    for (int d = 0; d < 3; ++d) {
        if (ext[d] < 1 || ext[d] > maxn) return false;
        if (largest_prime_factor((int)ext[d]) >= 29 && bluestein_lds((int)ext[d], 2 * esize) > 160 * 1024) return false;
    }
    return true;
}

// ---- Bluestein tables: FFT_M of the conjugate chirp, per (N, sign, precision, device); computed on first use by one
// workgroup and kept (at most a few hundred KB each, at most kBluMax tables: a full cache takes no more, further
// lengths run their direct-DFT stages) until lago_fluid_cache_clear() releases them.  The first use synchronises the stream once (another stream
// may be the next user); while THIS stream is being captured a missing table is not built and the direct-DFT stages
// serve that call; the allocation itself runs with the thread's capture mode relaxed, so that a global-mode capture in
// progress on another thread is not invalidated by it.
struct BluKey {
    int N, sign, esize, dev;
    bool operator<(const BluKey &o) const {
        return std::tie(N, sign, esize, dev) < std::tie(o.N, o.sign, o.esize, o.dev);
    }
};
// A table is held through a shared_ptr (as the coefficient tables of fft.hip): a lookup returns a reference that the
// caller keeps until its launch is enqueued, so lago_fluid_cache_clear() on another thread cannot free a table between a
// lookup and the launch that reads it -- the last reference frees it, and hipFree waits for the kernels already enqueued.
struct BluTab {
    void *d = nullptr;
    ~BluTab() {
        if (!d) return;
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;   // (a capture in progress on another thread is not invalidated)
        const bool swapped = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess;
        (void)hipFree(d);
        if (swapped) (void)hipThreadExchangeStreamCaptureMode(&mode);
    }
};
using BluRef = std::shared_ptr<BluTab>;
static std::mutex g_blu_mu;
static std::map<BluKey, BluRef> *g_blu = nullptr;   // (heap: never destroyed, as the coefficient cache of fft.hip)
constexpr size_t kBluMax = 64;
void bluestein_cache_clear() {
    std::map<BluKey, BluRef> dropped;   // released (and freed) after the lock: lookups are not stalled behind hipFree
    {
        std::lock_guard<std::mutex> lk(g_blu_mu);
        if (g_blu) dropped.swap(*g_blu);
    }
}
int bluestein_cache_entries() {
    std::lock_guard<std::mutex> lk(g_blu_mu);
    return g_blu ? (int)g_blu->size() : 0;
}

template <typename R>
static BluRef bluestein_table(int N, int M, int sign, hipStream_t s) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    const BluKey key{N, sign, (int)sizeof(R), dev};
    std::lock_guard<std::mutex> lk(g_blu_mu);
    if (!g_blu) g_blu = new std::map<BluKey, BluRef>();
    auto it = g_blu->find(key);
    if (it != g_blu->end()) return it->second;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
    const size_t cb = 2 * sizeof(R);
    // a full cache takes no further tables (the direct-DFT stages serve that length); lago_fluid_cache_clear() empties it
    if (g_blu->size() >= kBluMax) return nullptr;
    BluRef t = std::make_shared<BluTab>();
    {
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        const bool swapped = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess;
        const hipError_t e = hipMalloc(&t->d, (size_t)M * cb);
        if (swapped) (void)hipThreadExchangeStreamCaptureMode(&mode);
        if (e != hipSuccess) { (void)hipGetLastError(); t->d = nullptr; return nullptr; }
    }
    GLines a;
    a.N = N; a.M = M; a.sign = sign; a.L = 1; a.Lp = 1; a.mode = 0; a.inner = 1; a.nlines = 1; a.ppp = 1; a.chunks = 1;
    a.nhalf = N / 2 + 1; a.bhat = nullptr; a.inplace = 0;
    factorise_pow2(M, a);
    a.dinner = FastDiv(1u); a.dN = FastDiv((uint32_t)N); a.dL = FastDiv(1u); a.dnhalf = FastDiv((uint32_t)a.nhalf); a.dchunks = FastDiv(1u);
    for (int f = 0, st = 1; f < a.nfac; ++f) { a.ds[f] = FastDiv((uint32_t)st); a.dr[f] = FastDiv((uint32_t)(a.fac[f] + 1) / 2u); st *= a.fac[f]; }
    const size_t smem = (size_t)3 * M * cb;
    auto k = bluestein_table_kernel<R>;
    if (smem > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return nullptr;   // (~BluTab frees the buffer)
    hipLaunchKernelGGL(k, dim3(1), dim3(kBlock), smem, s, reinterpret_cast<GC<R> *>(t->d), a);
    if (hipStreamSynchronize(s) != hipSuccess) return nullptr;
    (*g_blu)[key] = t;
    return t;
}

// mode 0: `nlines` complex lines of N points, stride `inner`.  modes 1 / 2: `planes` fields of `nlines` contiguous real
// lines each (pairs of lines never cross a field: every batch item's result is independent of its neighbours)
template <typename R>
static int lines_pass(GC<R> *spec, const R *rin, R *rout, int N, uint64_t inner, uint64_t nlines, uint64_t planes, int sign,
                      int mode, hipStream_t s) {
    if (N == 1 && mode == 0) return LAGO_OK;
    if (nlines == 0 || planes == 0) return LAGO_OK;
    if (nlines >= (1ull << 31) || inner >= (1ull << 31) || planes * nlines >= (1ull << 31))
        return fail_invalid("fluid_metric: bad extent");
    GLines a;
    a.N = N;
    a.inner = (uint32_t)inner;
    a.nlines = (uint32_t)nlines;
    a.sign = sign;
    a.mode = mode;
    a.nhalf = N / 2 + 1;
    factorise(N, a);
    a.M = 0;
    a.bhat = nullptr;
    a.inplace = 0;
    const size_t cb = 2 * sizeof(R);
    const uint64_t units = mode == 0 ? nlines : (nlines + 1) / 2;
    // A prime factor r costs N r multiply-adds per line in the direct-DFT stage; from about 29 up the whole line is
    // cheaper as a Bluestein convolution (two power-of-two transforms of M >= 2 N - 1 points): 182 x 218 x 182 brain
    // volumes (218 = 2 * 109) 10.1 -> 3.6 ms per call, 193 x 229 x 193 22.1 -> 2.6 (profiles/r04_fft_generic.md).  Needs (2 (L | 1) + 1) M + N complex
    // numbers of LDS and the cached table; otherwise the direct stages serve.
    if (largest_prime_factor(N) >= 29) {   // (measured: 17 the same, 11 and 13 much slower than their direct stages)
        int M = 1;
        while (M < 2 * N - 1) M <<= 1;
        auto ldsb = [&](int l) { return (((size_t)2 * (l | 1) + 1) * M + N) * cb; };
        int L = std::max(1, 2048 / M);
        // (about 2048 points per workgroup and at least 32 B of adjacent lines on a strided pass: measured best of
        // 512 ... 4096 points x 16 ... 128 B)
        if (mode == 0) L = std::max(L, (int)(32 / cb));
        while (L > 1 && ldsb(L) > 96 * 1024) --L;
        if ((uint64_t)L > units) L = (int)units;
        // a strided pass: the stages in place (one line buffer), more adjacent lines per point for the same LDS
        auto ldsbi = [&](int l) { return (((size_t)(l | 1) + 1) * M + N) * cb; };
        bool inpl = false;
        // (64 B of adjacent lines: 182 x 218 x 182 3840 -> 3561 us; 128 B: 3735)
        if (mode == 0) {
            int Li = std::min((int)(64 / cb), 8192 / M);
            while (Li > 1 && ldsbi(Li) > 80 * 1024) --Li;
            if ((uint64_t)Li > units) Li = (int)units;
            if (Li > L) { L = Li; inpl = true; }
        }
        // (the reference lives until this function returns, i.e. until the launch that reads the table is enqueued)
        const BluRef tab = (inpl ? ldsbi(L) : ldsb(L)) <= 160 * 1024 ? bluestein_table<R>(N, M, sign, s) : nullptr;
        if (tab) {
            a.M = M;
            a.bhat = tab->d;
            a.inplace = inpl ? 1 : 0;
            factorise_pow2(M, a);
            const int Lp = L | 1;
            const size_t smem = inpl ? ldsbi(L) : ldsb(L);
            a.L = L;
            a.Lp = Lp;
            a.ppp = (uint32_t)units;
            a.chunks = (uint32_t)((units + L - 1) / L);
            a.dinner = FastDiv(a.inner);
            a.dN = FastDiv((uint32_t)N);
            a.dL = FastDiv((uint32_t)L);
            a.dnhalf = FastDiv((uint32_t)a.nhalf);
            a.dchunks = FastDiv(a.chunks);
            for (int f = 0, st = 1; f < a.nfac; ++f) {
                a.ds[f] = FastDiv((uint32_t)st);
                a.dr[f] = FastDiv((uint32_t)(a.fac[f] + 1) / 2u);
                st *= a.fac[f];
            }
            const uint64_t grid = mode == 0 ? a.chunks : planes * a.chunks;
            if (grid >= (1ull << 31)) return fail_invalid("fluid_metric: bad extent");
            auto k = inpl ? fft_lines_kernel<R, 4, true> : fft_lines_kernel<R, 4>;
            if (smem > 64 * 1024) LAGO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            hipLaunchKernelGGL(k, dim3((uint32_t)grid), dim3(kBlock), smem, s, spec, rin, rout, a);
            return LAGO_OK;
        }
    }
    // lines per workgroup: about 1024 points (a radix-4 stage is then one butterfly per thread, and the workgroup's LDS
    // stays small enough for 4+ workgroups per CU: measured best of 512 ... 4096, profiles/r04_fft_generic.md); a
    // strided pass takes at least 128 B of adjacent lines per point (64 B where 128 B would need more than half the
    // LDS); all within 128 KB
    auto lds = [&](int l) { return ((size_t)2 * (l | 1) + 1) * N * cb; };
    int L = std::max(1, 1024 / N);
    if (mode == 0) {
        const int want = std::max(L, (int)(128 / cb));
        L = lds(want) <= 80 * 1024 ? want : std::max(L, (int)(64 / cb));
    }
    while (L > 1 && lds(L) > 128 * 1024) --L;
    bool pow2 = true;   // (factors 8, 4 and 2 only: the instantiation without the odd radices)
    for (int f = 0; f < a.nfac; ++f) pow2 = pow2 && (a.fac[f] == 2 || a.fac[f] == 4 || a.fac[f] == 8);
    // long power-of-two lines of a strided pass (1024-point image rows: 123 KB for six lines in the ping-pong form, one
    // workgroup per CU): the stages run in place -- one line buffer, up to 8192 points in the registers of the workgroup
    if (pow2 && mode == 0 && lds(L) > 64 * 1024) {
        auto ldsi = [&](int l) { return ((size_t)(l | 1) + 1) * N * cb; };
        int Li = std::min((int)(128 / cb), 8192 / N);
        while (Li > 1 && ldsi(Li) > 80 * 1024) --Li;
        if (Li >= L) {
            L = Li;
            a.inplace = 1;
        }
    }
    if ((uint64_t)L > units) L = (int)units;
    const int Lp = L | 1;
    const size_t smem = a.inplace ? ((size_t)Lp + 1) * N * cb : ((size_t)2 * Lp + 1) * N * cb;
    if (smem > 160 * 1024) return fail_invalid("fluid_metric: extent %d is above what the generic FFT passes hold in LDS", N);
    a.L = L;
    a.Lp = Lp;
    a.ppp = (uint32_t)units;
    a.chunks = (uint32_t)((units + L - 1) / L);
    a.dinner = FastDiv(a.inner);
    a.dN = FastDiv((uint32_t)N);
    a.dL = FastDiv((uint32_t)L);
    a.dnhalf = FastDiv((uint32_t)a.nhalf);
    a.dchunks = FastDiv(a.chunks);
    for (int f = 0, st = 1; f < a.nfac; ++f) {
        a.ds[f] = FastDiv((uint32_t)st);
        a.dr[f] = FastDiv((uint32_t)(a.fac[f] + 1) / 2u);
        st *= a.fac[f];
    }
    const uint64_t grid = mode == 0 ? a.chunks : planes * a.chunks;
    if (grid >= (1ull << 31)) return fail_invalid("fluid_metric: bad extent");
    // a factor 11 or 13 beside powers of two only (176 = 16 * 11, 208 = 16 * 13): the instantiation that holds their
    // butterflies in registers too (152 / 161 VGPRs; with a radix-7 stage in the same line it loses: 91 = 7 * 13)
    bool r13 = false, odd_small = false;
    for (int f = 0; f < a.nfac; ++f) {
        r13 = r13 || a.fac[f] == 11 || a.fac[f] == 13;
        odd_small = odd_small || a.fac[f] == 3 || a.fac[f] == 5 || a.fac[f] == 7;
    }
    auto k = a.inplace ? fft_lines_kernel<R, 4, true>
             : pow2    ? fft_lines_kernel<R, 4>
             : (r13 && !odd_small) ? fft_lines_kernel<R, 13> : fft_lines_kernel<R, 7>;
    if (smem > 64 * 1024) LAGO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(k, dim3((uint32_t)grid), dim3(kBlock), smem, s, spec, rin, rout, a);
    return LAGO_OK;
}

std::atomic<int> g_generic_fuse{1};   // 1 (default): the x pass of the 3D generic path carries the operator (fft_xop_kernel)
void tune_generic_fuse(int on) { g_generic_fuse = on ? 1 : 0; }

// x forward + operator + x inverse in one launch; returns 1 when the length is left to the separate passes (Bluestein
// lines, lengths whose three-component chunk does not fit the LDS).
template <typename R>
static int xop_pass(GC<R> *spec, int inverse, const R *cosX, const R *sinX, const R *cosY, const R *sinY, const R *cosZ,
                    const R *sinZ, double alpha, double beta, double gamma, int dim, int64_t nn, int64_t X, int64_t Y, int64_t zc,
                    double scale, hipStream_t s) {
    // 3D: lines along X, columns (y, z-bin).  2D (geometry (1, nx, ny)): lines along Y = the fields' first axis, columns z-bin
    const int N = (int)(dim == 3 ? X : Y);
    if (dim == 2) Y = 1;
    if (N < 2 || largest_prime_factor(N) >= 29) return 1;
    {   // a factor 11 or 13 beside powers of two only takes the 155 - 164-register instantiation of the stages: with three
        // components' lines per workgroup on top it loses to the separate passes (182 x 218 x 182: +1.6 %)
        GLines t;
        factorise(N, t);
        bool r13 = false, odd_small = false;
        for (int f = 0; f < t.nfac; ++f) {
            r13 = r13 || t.fac[f] == 11 || t.fac[f] == 13;
            odd_small = odd_small || t.fac[f] == 3 || t.fac[f] == 5 || t.fac[f] == 7;
        }
        if (r13 && !odd_small) return 1;
    }
    const uint64_t cols = (uint64_t)Y * (uint64_t)zc;
    if (cols >= (1ull << 31) || (uint64_t)nn * cols >= (1ull << 31)) return 1;
    GLines a;
    a.N = N;
    a.inner = (uint32_t)cols;
    a.nlines = 0;
    a.sign = -1;
    a.mode = 0;
    a.nhalf = N / 2 + 1;
    factorise(N, a);
    a.M = 0;
    a.bhat = nullptr;
    a.inplace = 0;
    const size_t cb = 2 * sizeof(R);
    auto lds = [&](int lc) { return ((size_t)2 * ((dim * lc) | 1) + 1) * N * cb; };
    // columns per workgroup: FEW -- three (float64) or four (float32) adjacent columns, 48 / 32 B, within 40 KB of LDS, so
    // that four and more of these long-lived workgroups (load, forward stages, operator, inverse stages, store) share a CU.
    // Measured (tools/time_generic_fuse.py with the column count forced, profiles/r06_generic_fuse.md): float64 128^3
    // 2 / 3 / 4 / 6 / 8 columns -3.7 / -9.8 / -4.8 / -3.4 / +13 % against the separate passes, float64 160^3 -8.1 / -8.1 /
    // -7.0 / +13.8 / +5.9 %, float32 120^3 +3.9 / +0.8 / -9.5 / -7.9 / -7.4 %.
    int Lc = sizeof(R) == 8 ? 3 : 4;
    while (Lc > 1 && lds(Lc) > 40 * 1024) --Lc;
    // lines of 256 points and more, power-of-two: the stages IN PLACE (stage_inplace: one line buffer, half the LDS per column)
    // with up to 96 B of adjacent columns within 84 KB.  Measured against the ping-pong form above (tools/sweep_xop_inplace.py,
    // profiles/r06_xop_inplace.md): float64 256 x 128 x 128 -20 % (6 columns), float64 512^2 -24 % (4), float32 1024^2 -30 %
    // (4), float32 256^2 -12 % (12); 128- and 64-point lines gain nothing at any column count (+2 ... -4 %) and keep it.
    auto ldsi = [&](int lc) { return ((size_t)((dim * lc) | 1) + 1) * N * cb; };
    bool inpl = false;
    if (N >= 256) {
        bool p2 = true;
        for (int f = 0; f < a.nfac; ++f) p2 = p2 && (a.fac[f] == 2 || a.fac[f] == 4 || a.fac[f] == 8);
        int Li = (int)(96 / cb);
        while (Li > 1 && (dim * Li * N > 8192 || ldsi(Li) > 84 * 1024)) --Li;   // (stage_inplace: L N <= 8192)
        if (p2 && Li >= 2 && dim * Li * N <= 8192 && ldsi(Li) <= 84 * 1024) {
            Lc = Li;
            inpl = true;
        }
    }
    if ((inpl ? ldsi(Lc) : lds(Lc)) > 160 * 1024) return 1;
    if ((uint64_t)Lc > cols) Lc = (int)cols;
    a.L = dim * Lc;
    a.Lp = a.L | 1;
    a.ppp = 0;
    a.chunks = 0;
    a.dinner = FastDiv(a.inner);
    a.dN = FastDiv((uint32_t)N);
    a.dL = FastDiv((uint32_t)a.L);
    a.dnhalf = FastDiv((uint32_t)a.nhalf);
    a.dchunks = FastDiv(1u);
    bool pow2 = true, r13 = false, odd_small = false;
    for (int f = 0, st = 1; f < a.nfac; ++f) {
        a.ds[f] = FastDiv((uint32_t)st);
        a.dr[f] = FastDiv((uint32_t)(a.fac[f] + 1) / 2u);
        st *= a.fac[f];
        pow2 = pow2 && (a.fac[f] == 2 || a.fac[f] == 4 || a.fac[f] == 8);
        r13 = r13 || a.fac[f] == 11 || a.fac[f] == 13;
        odd_small = odd_small || a.fac[f] == 3 || a.fac[f] == 5 || a.fac[f] == 7;
    }
    XopArgs o;
    o.cosX = cosX; o.sinX = sinX; o.cosY = cosY; o.sinY = sinY; o.cosZ = cosZ; o.sinZ = sinZ;
    o.alpha = alpha; o.beta = beta; o.gamma = gamma; o.scale = scale;
    o.cols = (uint32_t)cols;
    o.zc = (uint32_t)zc;
    o.chunks = (uint32_t)((cols + Lc - 1) / Lc);
    o.dchunks = FastDiv(o.chunks);
    o.dzc = FastDiv(o.zc);
    o.dLc = FastDiv((uint32_t)Lc);
    o.Lc = Lc;
    const uint64_t grid = (uint64_t)nn * o.chunks;
    if (grid >= (1ull << 31)) return 1;
    const size_t smem = inpl ? ldsi(Lc) : lds(Lc);
    a.inplace = inpl ? 1 : 0;
#define LAGO_XOP_I()                                                                                                 \
    do {                                                                                                             \
        auto k = dim == 3 ? (inverse ? fft_xop_kernel<R, 4, true, 3, true> : fft_xop_kernel<R, 4, false, 3, true>)   \
                          : (inverse ? fft_xop_kernel<R, 4, true, 2, true> : fft_xop_kernel<R, 4, false, 2, true>);  \
        if (smem > 64 * 1024) LAGO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        hipLaunchKernelGGL(k, dim3((uint32_t)grid), dim3(kBlock), smem, s, spec, a, o);                              \
    } while (0)
#define LAGO_XOP(RM)                                                                                                 \
    do {                                                                                                             \
        auto k = dim == 3 ? (inverse ? fft_xop_kernel<R, RM, true, 3> : fft_xop_kernel<R, RM, false, 3>)             \
                          : (inverse ? fft_xop_kernel<R, RM, true, 2> : fft_xop_kernel<R, RM, false, 2>);            \
        if (smem > 64 * 1024) LAGO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        hipLaunchKernelGGL(k, dim3((uint32_t)grid), dim3(kBlock), smem, s, spec, a, o);                              \
    } while (0)
    if (pow2 && inpl) LAGO_XOP_I();
    else if (pow2) LAGO_XOP(4);
    else if (r13 && !odd_small) LAGO_XOP(13);
    else LAGO_XOP(7);
#undef LAGO_XOP
#undef LAGO_XOP_I
    return LAGO_OK;
}

// out = irfft(L^(+-2) rfft(m)) through the generic passes; work: the half spectrum (nn * dim * nx * ny * (nz/2 + 1) complex)
template <typename R>
int fluid_metric_generic(R *out, const R *m, R *work, int inverse, const R *cosX, const R *sinX, const R *cosY, const R *sinY,
                         const R *cosZ, const R *sinZ, double alpha, double beta, double gamma, int dim, int64_t nn, int64_t nx,
                         int64_t ny, int64_t nz, hipStream_t s) {
    // geometry order: 3D (X, Y, Z) = (nx, ny, nz); 2D (X, Y, Z) = (1, nx, ny)
    const int64_t X = dim == 3 ? nx : 1, Y = dim == 3 ? ny : nx, Z = dim == 3 ? nz : ny;
    const int64_t zc = Z / 2 + 1, planes = nn * dim;
    GC<R> *spec = reinterpret_cast<GC<R> *>(work);
    int rc = lines_pass<R>(spec, m, nullptr, (int)Z, 1, (uint64_t)(X * Y), (uint64_t)planes, -1, 1, s);
    if (rc != LAGO_OK) return rc;
    const double scale = 1.0 / ((double)X * (double)Y * (double)Z);
    // The last forward transform, the operator and the first inverse transform in one launch where the length allows
    // (fft_xop_kernel): along x in 3D, along y -- the fields' first axis -- in 2D.
    bool fused = false;
    if (dim == 3) {
        rc = lines_pass<R>(spec, nullptr, nullptr, (int)Y, (uint64_t)zc, (uint64_t)(planes * X * zc), 1, -1, 0, s);
        if (rc != LAGO_OK) return rc;
    }
    if (g_generic_fuse) {
        rc = xop_pass<R>(spec, inverse, cosX, sinX, cosY, sinY, cosZ, sinZ, alpha, beta, gamma, dim, nn, X, Y, zc, scale, s);
        if (rc < 0 || rc > 1) return rc;
        fused = rc == LAGO_OK;
    }
    if (!fused) {
        if (dim == 2) {
            rc = lines_pass<R>(spec, nullptr, nullptr, (int)Y, (uint64_t)zc, (uint64_t)(planes * X * zc), 1, -1, 0, s);
            if (rc != LAGO_OK) return rc;
        }
        rc = lines_pass<R>(spec, nullptr, nullptr, (int)X, (uint64_t)(Y * zc), (uint64_t)(planes * Y * zc), 1, -1, 0, s);
        if (rc != LAGO_OK) return rc;
        const int64_t cx = nx, cy = dim == 2 ? ny / 2 + 1 : ny, cz = dim == 3 ? nz / 2 + 1 : 1;
        rc = fluid_operator_impl<R>(work, inverse, cosX, sinX, cosY, sinY, cosZ, sinZ, alpha, beta, gamma, dim, nn, cx, cy, cz,
                                    (void *)s, scale);
        if (rc != LAGO_OK) return rc;
        rc = lines_pass<R>(spec, nullptr, nullptr, (int)X, (uint64_t)(Y * zc), (uint64_t)(planes * Y * zc), 1, +1, 0, s);
        if (rc != LAGO_OK) return rc;
        if (dim == 2) {
            rc = lines_pass<R>(spec, nullptr, nullptr, (int)Y, (uint64_t)zc, (uint64_t)(planes * X * zc), 1, +1, 0, s);
            if (rc != LAGO_OK) return rc;
        }
    }
    if (dim == 3) {
        rc = lines_pass<R>(spec, nullptr, nullptr, (int)Y, (uint64_t)zc, (uint64_t)(planes * X * zc), 1, +1, 0, s);
        if (rc != LAGO_OK) return rc;
    }
    rc = lines_pass<R>(spec, nullptr, out, (int)Z, 1, (uint64_t)(X * Y), (uint64_t)planes, +1, 2, s);
    if (rc != LAGO_OK) return rc;
    return finish_launch(s, "fluid_metric");
}

template int fluid_metric_generic<float>(float *, const float *, float *, int, const float *, const float *, const float *,
                                         const float *, const float *, const float *, double, double, double, int, int64_t,
                                         int64_t, int64_t, int64_t, hipStream_t);
template int fluid_metric_generic<double>(double *, const double *, double *, int, const double *, const double *,
                                          const double *, const double *, const double *, const double *, double, double,
                                          double, int, int64_t, int64_t, int64_t, int64_t, hipStream_t);

}  // namespace lago
