// LDS-resident FFT building blocks for the fluid metric (float32, extents 2^a, 3*2^a, 5*2^a), gfx950.
//
// FluidMetricOperator.forward of the reference (/root/reference/lagomorph/metric.py:11-19) is
// rfftn -> per-frequency 3x3 operator (cuda/metric.cu:163-305) -> irfftn.  Here it is three passes
// over HBM, each one workgroup-per-tile with the tile held in LDS:
//
//   zy forward : one (n, c, x) plane of nz reals x ny rows  ->  real-to-complex FFT along z (as an
//                nz/2-point complex FFT + split), complex FFT along y.  The two purely real
//                columns kz = 0 and kz = nz/2 ride through the y transform packed as one complex
//                column, so the plane is exactly ny x nz/2 complex = the size of the input.
//   x pass     : all nx and the three vector components of 16 neighbouring frequencies -> FFT
//                along x, operator, inverse FFT along x, in place.
//   zy inverse : mirror image of zy forward.
//
// Layout of the spectrum between the passes (the caller's `work` buffer, nn*3*nx*ny*(nz/2+1)
// complex): a "main" block [n][c][x][r][q < nz/2] followed by the Nyquist plane [n][c][x][r]
// (kz = nz/2), where row r holds ky = freq_at(r) and column q holds kz = freq_at(q): the
// digit-reversed order the transforms leave in LDS is kept in memory as well (the x pass is
// indifferent to it; its coefficient table is laid out the same way).  Splitting the odd 65th column off keeps every row a multiple of
// 128 bytes: measured on MI355X (tools/probes/seg_copy.hip), 128 B-aligned segments at a 64 KB
// stride stream at the speed of contiguous memory, while the 520 B rows of the usual
// nz/2+1 layout cost 3x on the write side (partial cache lines).
//
// All transforms are decimation-in-frequency forward (natural -> digit-reversed order) and
// decimation-in-time inverse (digit-reversed -> natural): one radix-3 or radix-5 level for the
// lengths 3*2^a / 5*2^a (96, 160, 192: the 160^3 volumes of BASELINE configs[4]), then radix-2 levels
// four at a time in registers (the stage plan below); the digit-reversed order is never undone, in LDS or in
// memory.
//
// Every phase between two workgroup barriers is a function of (phase, thread id) only, so the
// same code runs on the host with a loop over thread ids (tests/native/fft_emul.hip): the
// kernels are verified against a double-precision DFT without a GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lago {

// Cache policy of the passes' global traffic (read once / written once per pass).  LAGO_NT_* = 1 marks the access
// non-temporal (`nt`: streamed through the L2, evicted first) -- build-time switches for the A/B of
// profiles/r04_cache_policy.md; the defaults are what that measurement kept.
#ifndef LAGO_NT_X_LD
#define LAGO_NT_X_LD 0
#endif
#ifndef LAGO_NT_X_ST
#define LAGO_NT_X_ST 1
#endif
#ifndef LAGO_NT_ZF_LD
#define LAGO_NT_ZF_LD 0
#endif
#ifndef LAGO_NT_ZF_ST
#define LAGO_NT_ZF_ST 0
#endif
#ifndef LAGO_NT_ZI_LD
#define LAGO_NT_ZI_LD 0
#endif
#ifndef LAGO_NT_ZI_ST
#define LAGO_NT_ZI_ST 0
#endif
typedef float lg_f32x4_t __attribute__((ext_vector_type(4)));
template <int NT>
__host__ __device__ __forceinline__ float4 ldg4(const float4 *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (NT != 0) {
        const lg_f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const lg_f32x4_t *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
#endif
    return *p;
}
template <int NT>
__host__ __device__ __forceinline__ void stg4(float4 *p, float4 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (NT != 0) {
        lg_f32x4_t w;
        w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
        __builtin_nontemporal_store(w, reinterpret_cast<lg_f32x4_t *>(p));
        return;
    }
#endif
    *p = v;
}

namespace fl {

#define LAGO_HD __host__ __device__ __forceinline__

LAGO_HD float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}
LAGO_HD float2 cmulc(float2 a, float2 b) {  // a * conj(b)
    return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -(a.x * b.y)));
}
LAGO_HD int wave_uniform(int v) {  // v is the same in every lane of the wave: keep it (and what follows from it) on the scalar unit
#ifdef __HIP_DEVICE_COMPILE__
    return __builtin_amdgcn_readfirstlane(v);
#else
    return v;
#endif
}
LAGO_HD int brev(int v, int bits) { return bits ? (int)(__builtin_bitreverse32((uint32_t)v) >> (32 - bits)) : 0; }

// Roots of unity are taken in DOUBLE and rounded once (round 6): `sincospif` of the float-rounded argument 2 t / m is off by
// up to |arg| * 2^-24 * pi = 3.7e-7 (six float ulps near arg = 2), which showed as a 1.5 - 2 x larger float32 error of
// whole transforms than rocFFT's or pocketfft's (profiles/r06_twiddles.md).  The tables are built once per workgroup
// (one root per thread): the double evaluation costs nothing measurable.
LAGO_HD float2 twiddle(int t, int m) {  // exp(-2 pi i t / m)
#ifdef __HIP_DEVICE_COMPILE__
    double sn, cs;
    sincospi(-2.0 * (double)t / (double)m, &sn, &cs);
    return make_float2((float)cs, (float)sn);
#else
    const double a = -2.0 * 3.14159265358979323846 * (double)t / (double)m;
    return make_float2((float)cos(a), (float)sin(a));
#endif
}

constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int cgcd(int a, int b) { return b == 0 ? a : cgcd(b, a % b); }
constexpr int clcm(int a, int b) { return a / cgcd(a, b) * b; }

// A transform length N = R * 2^L2 with R odd, R <= 15 (64, 96, 128, 160, 192, 256, ...; round 6: 176 = 11 * 16 and
// 208 = 13 * 16 -- the 176 x 208 x 176 brain volumes of the OASIS images -- and 112 / 224, 144, 240 = 7, 9, 15 times 2^a).
// Forward = decimation in frequency: one radix-R level (R > 1), then the radix-2 levels of the R sub-transforms of
// 2^L2 points, up to four at a time in registers (stage plan below).  Frequency k = R*k2 + k1 ends up at position
// k1*2^L2 + bitrev(k2): the order is never undone, in LDS or in memory.  Inverse = the same data
// flow backwards (decimation in time, conjugated twiddles).
template <int R_, int L2_>
struct Sz {
    static constexpr int R = R_, L2 = L2_, M = 1 << L2_, N = R_ * (1 << L2_);
    static_assert(R_ == 1 || R_ == 3 || R_ == 5 || R_ == 7 || R_ == 9 || R_ == 11 || R_ == 13 || R_ == 15,
                  "an odd factor up to 15 times a power of two");
};
template <class S>
LAGO_HD int pos_of(int k) {  // position of frequency k after the forward transform
    if (S::R == 1) return brev(k, S::L2);
    return (k % S::R) * S::M + brev(k / S::R, S::L2);
}
template <class S>
LAGO_HD int freq_at(int p) {  // frequency held at position p
    if (S::R == 1) return brev(p, S::L2);
    return S::R * brev(p & (S::M - 1), S::L2) + (p >> S::L2);
}

// Stage plan of a transform N = R * 2^L2.  The radix-2 levels run in registers in groups of at most FOUR (16 points per
// work item), from the largest half down; where that saves a stage (L2 = 5: 96 = 3*32 and 160 = 5*32 points) the
// radix-R level takes the top radix-2 level with it (a radix-2R stage on 2R points).  80 = 5*16, 128 and 160 points are
// two stages, 64 two, 256 two -- every stage is one read and one write of the tile in LDS, which is what the passes
// are bound by once their HBM traffic is hidden (tools/probes/zy_probe.hip).
constexpr bool plan_fuse(int r, int l2) { return (r == 3 || r == 5) && l2 == 5; }
constexpr int plan_lr(int r, int l2) { return l2 - (plan_fuse(r, l2) ? 1 : 0); }      // radix-2 levels left to the groups
constexpr int group_count(int lr) { return (lr + 3) / 4; }
constexpr int group_s(int lr, int g) {       // levels of group g: as even as possible, the larger groups first
    const int n = group_count(lr);
    return g >= n ? 0 : lr / n + (g < lr % n ? 1 : 0);
}
constexpr int group_top(int lr, int g) {     // log2 of the largest half of group g
    int top = lr - 1;
    for (int k = 0; k < g; ++k) top -= group_s(lr, k);
    return top;
}
template <class S>
constexpr int stage_count() { return (S::R > 1 ? 1 : 0) + group_count(plan_lr(S::R, S::L2)); }

// a * exp(-2 pi i k / 16) (CONJ: a * exp(+2 pi i k / 16)), 0 <= k < 8; k is a constant after unrolling
template <bool CONJ>
LAGO_HD float2 mul_root16(float2 a, int k) {
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
    if (k == 0) return a;
    if (k == 4) return CONJ ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
    const float cs = k == 1 ? c1 : k == 2 ? h : k == 3 ? s1 : k == 5 ? -s1 : k == 6 ? -h : -c1;
    const float sn = k == 1 ? s1 : k == 2 ? h : k == 3 ? c1 : k == 5 ? c1 : k == 6 ? h : s1;
    return CONJ ? cmul(a, make_float2(cs, sn)) : cmulc(a, make_float2(cs, sn));
}

// A family of NB * NL transforms of S::N points in LDS: element i of transform (b, lane) is
// buf[b*BS + i*ES + lane*LS].  tw[t] = exp(-2 pi i t / LTW), 0 <= t < LTW, with S::N dividing LTW.
template <class S_, int ES_, int LS_, int NL_, int NB_, int BS_, int LTW_, int NT_>
struct Xf {
    using S = S_;
    static constexpr int ES = ES_, LS = LS_, NL = NL_, NB = NB_, BS = BS_, LTW = LTW_, NT = NT_;
    static_assert(LTW_ % S_::N == 0, "the twiddle table must hold the N-th roots of unity");
};

// Radix-2 group G of the 2^L2-point sub-transforms of family X: S = group_s levels (halves 2^TOP .. 2^(TOP-S+1)) on
// 2^S elements spaced 2^(TOP-S+1) apart, in registers.  FWD: decimation in frequency; !FWD: decimation in time
// with conjugated twiddles.  The twiddle of level l at element offset low + m' H is exp(-2 pi i low / 2^(lhalf+1))
// -- ONE table read per level -- times the constant exp(-2 pi i m' / 2^(S-l)), a 16th root of unity.
template <class X, int G, bool FWD>
LAGO_HD void radix2_stage(float2 *buf, const float2 *tw, int tid) {
    using Sq = typename X::S;
    constexpr int LR = plan_lr(Sq::R, Sq::L2), TOP = group_top(LR, G), S = group_s(LR, G);
    if constexpr (S > 0) {
        constexpr int M = Sq::M, R = 1 << S, LH = TOP - S + 1, H = 1 << LH;
        constexpr int PER = M / R, ITEMS = X::NB * Sq::R * PER * X::NL;
        for (int w = tid; w < ITEMS; w += X::NT) {
            const int lane = w % X::NL;
            const int q = w / X::NL;
            const int gidx = q & (PER - 1);
            const int bsub = q / PER;                      // (block, sub-transform)
            const int b = bsub / Sq::R, k1 = bsub % Sq::R;
            const int low = gidx & (H - 1);
            const int i = ((gidx >> LH) << (LH + S)) | low;
            float2 *p = buf + b * X::BS + (k1 * M + i) * X::ES + lane * X::LS;
            float2 v[R];
#pragma unroll
            for (int m = 0; m < R; ++m) v[m] = p[m * H * X::ES];
#pragma unroll
            for (int ll = 0; ll < S; ++ll) {
                const int l = FWD ? ll : S - 1 - ll;
                const int hm = R >> (l + 1);
                const int lhalf = LH + (S - 1 - l);
                // H == 1 (the last group: low = 0): the twiddles are the constants alone, no table read
                const float2 wl = H == 1 ? make_float2(1.f, 0.f) : tw[low * (X::LTW >> (lhalf + 1))];   // exp(-2 pi i low / 2^(lhalf+1))
#pragma unroll
                for (int mp = 0; mp < hm; ++mp) {
                    const int k16 = mp * (16 >> (S - l));
                    const float2 wv = H == 1 ? wl : mul_root16<false>(wl, k16);
#pragma unroll
                    for (int hi = 0; hi < (R >> 1) / hm; ++hi) {
                        const int m = hi * 2 * hm + mp;
                        const float2 a = v[m];
                        if (FWD) {
                            const float2 bb = v[m + hm];
                            const float2 d = make_float2(a.x - bb.x, a.y - bb.y);
                            v[m] = make_float2(a.x + bb.x, a.y + bb.y);
                            v[m + hm] = H == 1 ? mul_root16<false>(d, k16) : cmul(d, wv);
                        } else {
                            const float2 bb = H == 1 ? mul_root16<true>(v[m + hm], k16) : cmulc(v[m + hm], wv);
                            v[m] = make_float2(a.x + bb.x, a.y + bb.y);
                            v[m + hm] = make_float2(a.x - bb.x, a.y - bb.y);
                        }
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < R; ++m) p[m * H * X::ES] = v[m];
        }
    }
}

// cos / sin of 2 pi m / R for the odd radices 7 ... 15 (m is a constant after unrolling: the chains fold away)
template <int R>
LAGO_HD constexpr float odd_cos(int m) {
    const int h = m > R / 2 ? R - m : m;
    if (h == 0) return 1.0f;
    if (R == 9) return h == 1 ? 0.76604444311897801345f : h == 2 ? 0.17364817766693035894f : h == 3 ? -0.5f : -0.93969262078590842791f;
    if (R == 15)
        return h == 1 ? 0.91354545764260086660f : h == 2 ? 0.66913060635885823757f : h == 3 ? 0.30901699437494745126f
             : h == 4 ? -0.10452846326765347085f : h == 5 ? -0.5f : h == 6 ? -0.80901699437494745126f : -0.97814760073380568883f;
    if (R == 7) return h == 1 ? 0.62348980185873348336f : h == 2 ? -0.22252093395631439288f : -0.90096886790241914600f;
    if (R == 11)
        return h == 1 ? 0.84125353283118120551f : h == 2 ? 0.41541501300188643508f : h == 3 ? -0.14231483827328514358f
             : h == 4 ? -0.65486073394528510061f : -0.95949297361449736865f;
    return h == 1 ? 0.88545602565320991051f : h == 2 ? 0.56806474673115581187f : h == 3 ? 0.12053668025532304764f
         : h == 4 ? -0.35460488704253562142f : h == 5 ? -0.74851074817110108128f : -0.97094181742605201180f;
}
template <int R>
LAGO_HD constexpr float odd_sin(int m) {
    const int h = m > R / 2 ? R - m : m;
    const float sg = m > R / 2 ? -1.0f : 1.0f;
    if (h == 0) return 0.0f;
    if (R == 9)
        return sg * (h == 1 ? 0.64278760968653936292f : h == 2 ? 0.98480775301220802032f : h == 3 ? 0.86602540378443859659f
                   : 0.34202014332566871291f);
    if (R == 15)
        return sg * (h == 1 ? 0.40673664307580020827f : h == 2 ? 0.74314482547739424412f : h == 3 ? 0.95105651629515353118f
                   : h == 4 ? 0.99452189536827328986f : h == 5 ? 0.86602540378443859659f : h == 6 ? 0.58778525229247313710f
                   : 0.20791169081775934258f);
    if (R == 7) return sg * (h == 1 ? 0.78183148246802980363f : h == 2 ? 0.97492791218182361934f : 0.43388373911755812040f);
    if (R == 11)
        return sg * (h == 1 ? 0.54064081745559755543f : h == 2 ? 0.90963199535451833011f : h == 3 ? 0.98982144188093268422f
                   : h == 4 ? 0.75574957435425826890f : 0.28173255684142967104f);
    return sg * (h == 1 ? 0.46472317204376856203f : h == 2 ? 0.82298386589365635224f : h == 3 ? 0.99270887409805397272f
               : h == 4 ? 0.93501624268541483342f : h == 5 ? 0.66312265824079519305f : 0.23931566428755776665f);
}
// R-point DFT of v[0..R) in place; SGN = -1 forward (exp(-2 pi i r k / R)), +1 inverse
template <int R, int SGN>
LAGO_HD void dft_small(float2 *v) {
    if constexpr (R == 3) {
        const float s = 0.86602540378443864676f;  // sin(2 pi / 3)
        const float2 a = make_float2(v[1].x + v[2].x, v[1].y + v[2].y), d = make_float2(v[1].x - v[2].x, v[1].y - v[2].y);
        const float2 c = make_float2(fmaf(-0.5f, a.x, v[0].x), fmaf(-0.5f, a.y, v[0].y));
        const float2 j = SGN < 0 ? make_float2(s * d.y, -s * d.x) : make_float2(-s * d.y, s * d.x);  // -+ i s d
        v[0] = make_float2(v[0].x + a.x, v[0].y + a.y);
        v[1] = make_float2(c.x + j.x, c.y + j.y);
        v[2] = make_float2(c.x - j.x, c.y - j.y);
    } else if constexpr (R == 5) {
        const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;   // cos(2 pi/5), cos(4 pi/5)
        const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;    // sin(2 pi/5), sin(4 pi/5)
        const float2 a1 = make_float2(v[1].x + v[4].x, v[1].y + v[4].y), b1 = make_float2(v[1].x - v[4].x, v[1].y - v[4].y);
        const float2 a2 = make_float2(v[2].x + v[3].x, v[2].y + v[3].y), b2 = make_float2(v[2].x - v[3].x, v[2].y - v[3].y);
        const float2 e1 = make_float2(fmaf(c2, a2.x, fmaf(c1, a1.x, v[0].x)), fmaf(c2, a2.y, fmaf(c1, a1.y, v[0].y)));
        const float2 e2 = make_float2(fmaf(c1, a2.x, fmaf(c2, a1.x, v[0].x)), fmaf(c1, a2.y, fmaf(c2, a1.y, v[0].y)));
        const float2 o1 = make_float2(fmaf(s2, b2.x, s1 * b1.x), fmaf(s2, b2.y, s1 * b1.y));
        const float2 o2 = make_float2(fmaf(-s1, b2.x, s2 * b1.x), fmaf(-s1, b2.y, s2 * b1.y));
        // forward: y_k = e -+ i o  (k = 1, 2 take -i o; k = 4, 3 take +i o); inverse: the signs swap
        const float2 j1 = SGN < 0 ? make_float2(o1.y, -o1.x) : make_float2(-o1.y, o1.x);
        const float2 j2 = SGN < 0 ? make_float2(o2.y, -o2.x) : make_float2(-o2.y, o2.x);
        v[0] = make_float2(v[0].x + a1.x + a2.x, v[0].y + a1.y + a2.y);
        v[1] = make_float2(e1.x + j1.x, e1.y + j1.y);
        v[4] = make_float2(e1.x - j1.x, e1.y - j1.y);
        v[2] = make_float2(e2.x + j2.x, e2.y + j2.y);
        v[3] = make_float2(e2.x - j2.x, e2.y - j2.y);
    }
}

// exp(-2 pi i k / (2 R)), 0 < k < R: the twiddle of the radix-R level half a sub-transform further on
template <int R>
LAGO_HD float2 root_2r(int k) {
    if (R == 3) {   // 6th roots
        return k == 1 ? make_float2(0.5f, -0.86602540378443864676f) : make_float2(-0.5f, -0.86602540378443864676f);
    }
    // 10th roots
    return k == 1 ? make_float2(0.80901699437494742410f, -0.58778525229247312917f)
         : k == 2 ? make_float2(0.30901699437494742410f, -0.95105651629515357212f)
         : k == 3 ? make_float2(-0.30901699437494742410f, -0.95105651629515357212f)
                  : make_float2(-0.80901699437494742410f, -0.58778525229247312917f);
}

// The radix-R level: x[m + M r] -> y[k1][m] = (sum_r x[m + M r] W_R^(r k1)) W_N^(m k1), stored at k1*M + m
// (forward); the inverse undoes it.  FUSED (plan_fuse): one work item takes m and m + M/2 and also runs the top
// radix-2 level of the M-point sub-transforms on them, y'[k1][m] = A + B, y'[k1][m + M/2] = (A - B) W_M^m; the
// twiddles of the second half are those of the first times the constants exp(-2 pi i k1 / 2R).
template <class X, bool FWD>
LAGO_HD void radixR_stage(float2 *buf, const float2 *tw, int tid) {
    using Sq = typename X::S;
    if constexpr (Sq::R > 1) {
        constexpr bool FUSED = plan_fuse(Sq::R, Sq::L2);
        constexpr int R = Sq::R, M = Sq::M, MW = FUSED ? M / 2 : M, ITEMS = X::NB * MW * X::NL, TWS = X::LTW / Sq::N;
        for (int w = tid; w < ITEMS; w += X::NT) {
            const int lane = w % X::NL;
            const int q = w / X::NL;
            const int m = q % MW, b = q / MW;
            float2 *p = buf + b * X::BS + m * X::ES + lane * X::LS;
            float2 v[R], u[FUSED ? R : 1], wk[R];
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = p[r * M * X::ES];
            if (FUSED) {
#pragma unroll
                for (int r = 0; r < R; ++r) u[r] = p[(r * M + M / 2) * X::ES];
            }
#pragma unroll
            for (int k1 = 1; k1 < R; ++k1) wk[k1] = tw[m * k1 * TWS];
            const float2 wm = FUSED ? tw[m * R * TWS] : make_float2(1.f, 0.f);   // W_M^m
            if (FWD) {
                dft_small<R, -1>(v);
#pragma unroll
                for (int k1 = 1; k1 < R; ++k1) v[k1] = cmul(v[k1], wk[k1]);
                if (FUSED) {
                    dft_small<R, -1>(u);
#pragma unroll
                    for (int k1 = 1; k1 < R; ++k1) u[k1] = cmul(u[k1], cmul(wk[k1], root_2r<R>(k1)));
#pragma unroll
                    for (int k1 = 0; k1 < R; ++k1) {
                        const float2 a = v[k1], bb = u[k1];
                        v[k1] = make_float2(a.x + bb.x, a.y + bb.y);
                        u[k1] = cmul(make_float2(a.x - bb.x, a.y - bb.y), wm);
                    }
                }
            } else {
                if (FUSED) {
#pragma unroll
                    for (int k1 = 0; k1 < R; ++k1) {
                        const float2 a = v[k1], bb = cmulc(u[k1], wm);
                        v[k1] = make_float2(a.x + bb.x, a.y + bb.y);
                        u[k1] = make_float2(a.x - bb.x, a.y - bb.y);
                    }
#pragma unroll
                    for (int k1 = 1; k1 < R; ++k1) u[k1] = cmulc(u[k1], cmul(wk[k1], root_2r<R>(k1)));
                    dft_small<R, +1>(u);
                }
#pragma unroll
                for (int k1 = 1; k1 < R; ++k1) v[k1] = cmulc(v[k1], wk[k1]);
                dft_small<R, +1>(v);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) p[r * M * X::ES] = v[r];
            if (FUSED) {
#pragma unroll
                for (int r = 0; r < R; ++r) p[(r * M + M / 2) * X::ES] = u[r];
            }
        }
    }
}

// The radix-R level for odd R = 7 ... 15, prime or not (never fused with a radix-2 level), through the symmetric pairs
// a_j = x_j + x_(R-j), d_j = x_j - x_(R-j): y_k = x_0 + sum_j cos(2 pi j k / R) a_j -+ i sum_j sin(2 pi j k / R) d_j and
// y_(R-k) its mirror, (R - 1)^2 multiply-adds.  STREAMED -- the inverse's twiddles are multiplied in as the inputs are read
// into the symmetric pairs, every output pair is multiplied by its (forward) twiddle and stored as soon as it is formed --
// so that a work item holds R - 1 pair
// values and one output pair instead of R inputs, R twiddles and R outputs: the 1024-thread persistent zy kernels of the
// 208 x 176 planes (128 registers per thread, nine float4 of the next plane in flight) spilled 12 - 23 registers with the
// array form (inputs, twiddles and outputs in arrays around an R-point DFT, as radixR_stage has it for R = 3, 5).
template <class X, bool FWD>
LAGO_HD void radix_odd_stage(float2 *buf, const float2 *tw, int tid) {
    using Sq = typename X::S;
    constexpr int R = Sq::R, M = Sq::M, H = (R - 1) / 2, ITEMS = X::NB * M * X::NL, TWS = X::LTW / Sq::N;
    for (int w = tid; w < ITEMS; w += X::NT) {
        const int lane = w % X::NL;
        const int q = w / X::NL;
        const int m = q % M, b = q / M;
        float2 *p = buf + b * X::BS + m * X::ES + lane * X::LS;
        const float2 x0 = p[0];
        float2 a[H], d[H];
#pragma unroll
        for (int j = 1; j <= H; ++j) {
            float2 u = p[j * M * X::ES], v = p[(R - j) * M * X::ES];
            if (!FWD) {
                u = cmulc(u, tw[m * j * TWS]);
                v = cmulc(v, tw[m * (R - j) * TWS]);
            }
            a[j - 1] = make_float2(u.x + v.x, u.y + v.y);
            d[j - 1] = make_float2(u.x - v.x, u.y - v.y);
        }
        float2 s = x0;
#pragma unroll
        for (int j = 0; j < H; ++j) s = make_float2(s.x + a[j].x, s.y + a[j].y);
        p[0] = s;
#pragma unroll
        for (int k = 1; k <= H; ++k) {
            float2 e = x0, o = make_float2(0.f, 0.f);
#pragma unroll
            for (int j = 1; j <= H; ++j) {
                const int mm = (j * k) % R;
                const float c = odd_cos<R>(mm), sn = odd_sin<R>(mm);
                e = make_float2(fmaf(c, a[j - 1].x, e.x), fmaf(c, a[j - 1].y, e.y));
                o = make_float2(fmaf(sn, d[j - 1].x, o.x), fmaf(sn, d[j - 1].y, o.y));
            }
            const float2 jv = FWD ? make_float2(o.y, -o.x) : make_float2(-o.y, o.x);   // -+ i o
            float2 yk = make_float2(e.x + jv.x, e.y + jv.y), ym = make_float2(e.x - jv.x, e.y - jv.y);
            if (FWD) {
                yk = cmul(yk, tw[m * k * TWS]);
                ym = cmul(ym, tw[m * (R - k) * TWS]);
            }
            p[k * M * X::ES] = yk;
            p[(R - k) * M * X::ES] = ym;
        }
    }
}

// stage number `g` of the forward transform / of the inverse transform (which runs the stages in reverse order);
// g is a constant after unrolling
template <class X, bool FWD>
LAGO_HD void run_stage(int g, float2 *buf, const float2 *tw, int tid) {
    using Sq = typename X::S;
    constexpr int G = stage_count<Sq>(), HASR = Sq::R > 1 ? 1 : 0;
    const int gg = FWD ? g : G - 1 - g;
    if (HASR && gg == 0) {
        if constexpr (Sq::R > 5) radix_odd_stage<X, FWD>(buf, tw, tid);
        else radixR_stage<X, FWD>(buf, tw, tid);
    }
    else if (gg - HASR == 0) radix2_stage<X, 0, FWD>(buf, tw, tid);
    else if (gg - HASR == 1) radix2_stage<X, 1, FWD>(buf, tw, tid);
    else radix2_stage<X, 2, FWD>(buf, tw, tid);
}

// ---------------------------------------------------------------------------------------------
// The per-frequency operator on one frequency bin (three complex components), coefficients as
// tabulated by fluid_coef_kernel: cuda/metric.cu:103-130 (sharp: Cholesky solve with
// ooG00 G10 ooG11 G20 G21 ooG22) and :145-160 (flat: L00 L10 L11 L20 L21 L22).  Same expressions
// and roundings as fluid_kernel in metric.hip.
template <bool INV>
LAGO_HD void fluid_bin(const float *c, float2 &X, float2 &Y, float2 &Z, float scale) {
    const float c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c4 = c[4], c5 = c[5];
    float bx[2] = {X.x, X.y}, by[2] = {Y.x, Y.y}, bz[2] = {Z.x, Z.y};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        float bX = bx[q], bY = by[q], bZ = bz[q];
        if (INV) {
            const float y0 = bX * c0;
            const float y1 = fmaf(-c1, y0, bY) * c2;
            const float y2 = fmaf(-c4, y1, fmaf(-c3, y0, bZ)) * c5;
            bZ = y2 * c5;
            bY = fmaf(-c4, bZ, y1) * c2;
            bX = fmaf(-c3, bZ, fmaf(-c1, bY, y0)) * c0;
        } else {
            const float x = fmaf(c3, bZ, fmaf(c0, bX, c1 * bY));
            const float yy = fmaf(c4, bZ, fmaf(c1, bX, c2 * bY));
            bZ = fmaf(c5, bZ, fmaf(c3, bX, c4 * bY));
            bX = x;
            bY = yy;
        }
        bx[q] = bX * scale; by[q] = bY * scale; bz[q] = bZ * scale;
    }
    X = make_float2(bx[0], bx[1]);
    Y = make_float2(by[0], by[1]);
    Z = make_float2(bz[0], bz[1]);
}

// ---------------------------------------------------------------------------------------------
// x pass.  Workgroup tile: 3 components x NX x 16 neighbouring bins (16 kz of one ky in the main
// block, or 16 ky of the Nyquist plane); thread = (row group, bin lane).
struct XArgs {
    float2 *main_, *nyq;        // split spectrum (see the header comment)
    const float *tabM, *tabN;   // coefficients [kx][r][q < nzh][6] and [kx][r][6], (r, q) as in the spectrum
    int ny, nzh, nch, items_per_n;   // nch = ny * nzh / 16 tiles of the main block, items_per_n = nch + ceil(ny / 16)
    int nn, ipw;                // batch size; batch items one workgroup runs through with the same coefficients
    float scale;
    uint32_t total;             // ceil(nn / ipw) * items_per_n workgroups
    int rev;                    // launch direction (common.hpp)
};

template <class SX, bool INV, int NT = 256>
struct XPass {
    static constexpr int NX = SX::N, KL = 16, KCP = KL + 1;  // odd row pitch spreads rows over banks
    using T = Xf<SX, KCP, 1, KL, 3, NX * KCP, NX, NT>;
    static constexpr int G = stage_count<SX>();
    static constexpr int NPH = 2 * G + 3;  // load | G forward stages | operator | G inverse stages | store
    static constexpr int ROWS_IT = NT / 8, KLD = (3 * NX + ROWS_IT - 1) / ROWS_IT;  // float4 loads: 8 lanes per 16-bin row
    static constexpr bool RAGGED = 3 * NX % ROWS_IT != 0;           // (176 and 208 points: the last pass is guarded)
    static constexpr int RG = NT / KL, NOP = (NX + RG - 1) / RG;    // operator: NOP bins per thread
    static constexpr bool RAGGED_OP = NX % RG != 0;                 // (208 points over 512 threads: 6 1/2 rows per thread)
    static constexpr size_t SMEM = (size_t)(3 * NX * KCP + NX) * sizeof(float2);
    LAGO_HD static bool row_ok(int rg, int k) { return !RAGGED || rg + k * ROWS_IT < 3 * NX; }
    LAGO_HD static bool op_ok(int p) { return !RAGGED_OP || p < NX; }
    // TAIL: the instantiations for NX = 88, 104, 120 (8 x odd) also serve planes with ny % 16 = 8, whose Nyquist plane ends
    // with half a tile (Block::nv = 8: the other lanes load zeros and store nothing).  The lengths with NX % 16 = 0 carry no
    // such code (their kernels sit at the register count that lets two workgroups share a CU) and take ny % 16 = 0 only.
    static constexpr bool TAIL = NX % 16 != 0;
    template <class B> LAGO_HD static bool bin_ok(const B &b, int c) { if constexpr (TAIL) return c < b.nv; else return true; }

    struct Block {  // workgroup-uniform
        float2 *base;
        size_t xs;        // x stride (complex elements); component stride = NX * xs
        const float *tb;  // coefficient row of bin lane 0 at kx = 0
        size_t tks;       // kx stride of the coefficient table (floats)
        int nv;           // bins of the tile that exist: 16, or ny % 16 (even) in the last tile of a Nyquist plane whose rows
                          // are not whole tiles (ny = 88, 104, 120): the other lanes load zeros and store nothing
    };
    struct Regs { float coef[NOP][6]; };

    LAGO_HD static Block locate(const XArgs &a, uint32_t blk) {
        return locate(a, blk / (uint32_t)a.items_per_n * (uint32_t)a.ipw, blk % (uint32_t)a.items_per_n);
    }
    LAGO_HD static Block locate(const XArgs &a, uint32_t n, uint32_t item) {   // batch item n, bin tile `item`
        // the (r, q) positions of one x are contiguous in memory: a tile is 16 consecutive positions of that run (16 kz of
        // one ky where nzh is a multiple of 16; the tiles of an 88-bin row straddle rows -- the pass does not care which
        // bins it holds, the coefficient table is laid out the same way)
        const uint32_t nmain = (uint32_t)a.nch;
        Block b;
        if (item < nmain) {
            b.xs = (size_t)a.ny * a.nzh;
            b.base = a.main_ + (size_t)n * 3 * NX * b.xs + (size_t)item * KL;
            b.tb = a.tabM + (size_t)item * KL * 6;
            b.tks = b.xs * 6;
            b.nv = KL;
        } else {
            const uint32_t j = item - nmain;
            b.xs = (size_t)a.ny;
            b.base = a.nyq + (size_t)n * 3 * NX * b.xs + j * KL;
            b.tb = a.tabN + (size_t)j * KL * 6;
            b.tks = b.xs * 6;
            b.nv = TAIL ? min(KL, a.ny - (int)j * KL) : KL;
        }
        return b;
    }

    // The load phase in pieces -- tile into registers, coefficients into registers, registers into LDS -- so that a
    // persistent workgroup can request its next tile while it transforms the current one (fft3.hip).
    LAGO_HD static void load_one(int tid, const Block &b, float4 (&v)[KLD], int k) {
        const int rg = tid >> 3, l8 = tid & 7;
        if (!bin_ok(b, 2 * l8)) v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        else if (row_ok(rg, k)) v[k] = ldg4<LAGO_NT_X_LD>(reinterpret_cast<const float4 *>(b.base + (size_t)(rg + k * ROWS_IT) * b.xs + 2 * l8));
    }
    LAGO_HD static void load_coef(int tid, Regs &r, const Block &b) {
        const int kc = tid & (KL - 1), row0 = tid / KL;
#pragma unroll
        for (int i = 0; i < NOP; ++i) {
            if (!op_ok(row0 + i * RG)) continue;
            if (!bin_ok(b, kc)) {   // (no such bin: zeros in, zeros out)
#pragma unroll
                for (int e = 0; e < 6; ++e) r.coef[i][e] = 0.f;
                continue;
            }
            const float *t = b.tb + (size_t)freq_at<SX>(row0 + i * RG) * b.tks + kc * 6;
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const float2 c2 = *reinterpret_cast<const float2 *>(t + 2 * e);
                r.coef[i][2 * e] = c2.x;
                r.coef[i][2 * e + 1] = c2.y;
            }
        }
    }
    LAGO_HD static void fill(int tid, const float4 (&v)[KLD], float2 *buf) {
        const int rg = tid >> 3, l8 = tid & 7;
#pragma unroll
        for (int k = 0; k < KLD; ++k) {
            if (!row_ok(rg, k)) continue;
            float2 *d = buf + (rg + k * ROWS_IT) * KCP + 2 * l8;
            d[0] = make_float2(v[k].x, v[k].y);
            d[1] = make_float2(v[k].z, v[k].w);
        }
    }
    LAGO_HD static void fill_twiddles(int tid, float2 *tw) {
        for (int t = tid; t < NX; t += NT) tw[t] = twiddle(t, NX);
    }

    // `first`: the workgroup's first batch item -- twiddles and coefficients are loaded then and kept
    LAGO_HD static void phase(int ph, int tid, Regs &r, const Block &b, float2 *buf, float2 *tw, float scale,
                              bool first = true) {
        if (ph == 0) {
            if (first) fill_twiddles(tid, tw);
            float4 v[KLD];
#pragma unroll
            for (int k = 0; k < KLD; ++k) load_one(tid, b, v, k);
            // coefficient prefetch for the operator phase (consumed after the forward stages)
            if (first) load_coef(tid, r, b);
            fill(tid, v, buf);
        } else if (ph <= G) {
            run_stage<T, true>(ph - 1, buf, tw, tid);
        } else if (ph == G + 1) {
            // position p of the x axis holds kx = freq_at(p) after the forward stages
            const int kc = tid & (KL - 1), row0 = tid / KL;
#pragma unroll
            for (int i = 0; i < NOP; ++i) {
                const int p = row0 + i * RG;
                if (!op_ok(p)) continue;
                float2 *bx = buf + (0 * NX + p) * KCP + kc, *by = buf + (1 * NX + p) * KCP + kc,
                       *bz = buf + (2 * NX + p) * KCP + kc;
                float2 X = *bx, Y = *by, Z = *bz;
                fluid_bin<INV>(r.coef[i], X, Y, Z, scale);
                *bx = X; *by = Y; *bz = Z;
            }
        } else if (ph <= 2 * G + 1) {
            run_stage<T, false>(ph - G - 2, buf, tw, tid);
        } else {
            const int rg = tid >> 3, l8 = tid & 7;
#pragma unroll
            for (int k = 0; k < KLD; ++k) {
                if (!row_ok(rg, k) || !bin_ok(b, 2 * l8)) continue;
                const float2 *s = buf + (rg + k * ROWS_IT) * KCP + 2 * l8;
                const float2 a = s[0], c = s[1];
                stg4<LAGO_NT_X_ST>(reinterpret_cast<float4 *>(b.base + (size_t)(rg + k * ROWS_IT) * b.xs + 2 * l8),
                                   make_float4(a.x, a.y, c.x, c.y));
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------
// zy passes.  LDS plane P[y][PZ], PZ = NZ/2 + 1 (odd pitch: column walks with lanes along y are
// conflict-free; the spare column holds the Nyquist bins around the store / load).
struct ZYArgs {
    const float *in;   // forward: real planes (nn*3*nx planes of ny*nz); inverse: unused
    float *out;        // inverse: real planes
    float2 *main_, *nyq;
    uint32_t total;
    int rev;           // launch direction (common.hpp): planes descending
    float oscale = 1.0f;   // inverse: factor on every stored value (lago_fluid_metric_scaled)
};

// threads per plane: 1024 for the planes that are alone on their CU (above 80 KB of LDS: the persistent kernels of
// fft3.hip, 128 VGPRs per thread), 512 for the others -- two or more workgroups share a CU and a four-level group is 16
// points per work item: 512 items at 128 x 64 complex, and more than the 64 VGPRs a 1024-thread pair could have; the
// load / store phases guard the last pass when the plane does not divide (160 x 80 / 2 = 6400 float4 over 1024 threads)
constexpr int zy_threads(int ny, int nzh) { return (size_t)ny * (nzh + 1) * 8 > 80 * 1024 ? 1024 : 512; }

template <class SY, class SZH, int NT = zy_threads(SY::N, SZH::N)>
struct ZY {
    static constexpr int NY = SY::N, NZH = SZH::N, NZ = 2 * NZH, PZ = NZH + 1;
    // one table of lcm(NY, NZ)-th roots serves both axes and the real-FFT split; where that table would push the plane
    // over the 160 KB of LDS (208 x 176: lcm 2288) the axes take a table each, NZ-th roots followed by NY-th roots
    static constexpr bool TW2 = (size_t)(NY * PZ + clcm(NY, NZ)) * sizeof(float2) > 160 * 1024;
    static constexpr int LTW = TW2 ? NZ : clcm(NY, NZ), LTWY = TW2 ? NY : LTW, TWY0 = TW2 ? NZ : 0, TWN = TW2 ? NZ + NY : LTW;
    using TZ = Xf<SZH, 1, PZ, NY, 1, 0, LTW, NT>;   // along z, lanes over y
    using TY = Xf<SY, PZ, 1, NZH, 1, 0, LTWY, NT>;  // along y, lanes over kz
    static constexpr int GZ = stage_count<SZH>(), GY = stage_count<SY>();
    static constexpr int NPH = GZ + GY + 4;  // load | stages | split | stages | unpack | store   (mirrored for the inverse)
    static constexpr int F4 = NY * NZH / 2;        // float4 (two complex) per plane
    static constexpr int KV = (F4 + NT - 1) / NT;  // ... per thread (the last pass is guarded)
    static constexpr size_t SMEM = (size_t)(NY * PZ + TWN) * sizeof(float2);
    static constexpr int THREADS = NT;
    static_assert(KV >= 1 && NZH % 2 == 0, "rows must hold whole float4");

    // Work split of the real-FFT split / merge phases: a wave takes 64 consecutive rows y (lanes) and every SLOTS-th k
    // from k0 on, so k is uniform in the wave.
    struct RowsK {
        static constexpr int CH = (NY + 63) / 64, SLOTS = (NT / 64) / CH;
        static_assert(SLOTS >= 1, "fewer waves than 64-row chunks");
        int y, k0;
        bool active;
        LAGO_HD RowsK(int tid) {
            const int wave = wave_uniform(tid >> 6);
            k0 = wave / CH;
            y = (wave % CH) * 64 + (tid & 63);
            active = k0 < SLOTS && y < NY;
        }
    };

    LAGO_HD static void fill_twiddles(int tid, float2 *tw) {
        for (int t = tid; t < LTW; t += NT) tw[t] = twiddle(t, LTW);
        if (TW2) for (int t = tid; t < LTWY; t += NT) tw[TWY0 + t] = twiddle(t, LTWY);
    }
    // phase 0 in two halves -- the plane's global loads into registers, registers into the LDS plane -- so that a
    // persistent workgroup can request its next plane while it transforms the current one (fft3.hip)
    LAGO_HD static void fwd_load(int tid, const float *in, float4 (&v)[KV]) {
#pragma unroll
        for (int k = 0; k < KV; ++k)
            if (tid + k * NT < F4) v[k] = ldg4<LAGO_NT_ZF_LD>(reinterpret_cast<const float4 *>(in) + (tid + k * NT));
    }
    LAGO_HD static void fwd_fill(int tid, const float4 (&v)[KV], float2 *P) {
        // a row of NZ reals is NZH complex z[j] = (x[2j], x[2j+1]) as it lies in memory
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            if (tid + k * NT >= F4) continue;
            const int e = (tid + k * NT) * 2, y = e / NZH, j = e % NZH;
            P[y * PZ + j] = make_float2(v[k].x, v[k].y);
            P[y * PZ + j + 1] = make_float2(v[k].z, v[k].w);
        }
    }
    // Column 0 of the inverse carries FA + i FB (FB = the Nyquist column): one thread per row r fetches both bins of
    // the row, c0 = (main[r][0], nyq[r]), and writes the packed value; the float4 owner of (r, 0..1) leaves it alone.
    // c0 rides in the LAST float4 slot, which only the first F4 % NT threads need for the plane itself: threads
    // C0T .. C0T + NY - 1 use theirs for c0 (no extra registers: the persistent kernels of fft3.hip sit at the 128 a
    // 1024-thread workgroup may have); where the plane fills the last slot completely, one more slot is appended.
    static constexpr bool C0_SHARES = F4 % NT != 0 && F4 % NT + NY <= NT;
    static constexpr int KVX = KV + (C0_SHARES ? 0 : 1), C0T = C0_SHARES ? F4 % NT : 0;
    LAGO_HD static void inv_load_c0(int tid, const float2 *mainp, const float2 *nyqp, float4 (&v)[KVX]) {
        if (tid >= C0T && tid < C0T + NY) {
            const float2 a = mainp[(tid - C0T) * NZH], b = nyqp[tid - C0T];
            v[KVX - 1] = make_float4(a.x, a.y, b.x, b.y);
        }
    }
    LAGO_HD static void inv_load(int tid, const float2 *mainp, const float2 *nyqp, float4 (&v)[KVX]) {
#pragma unroll
        for (int k = 0; k < KV; ++k)
            if (tid + k * NT < F4) v[k] = ldg4<LAGO_NT_ZI_LD>(reinterpret_cast<const float4 *>(mainp) + (tid + k * NT));
        inv_load_c0(tid, mainp, nyqp, v);
    }
    LAGO_HD static void inv_fill(int tid, const float4 (&v)[KVX], float2 *P) {
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            if (tid + k * NT >= F4) continue;
            const int e = (tid + k * NT) * 2, r = e / NZH, c = e % NZH;   // the spectrum lies in memory as it lies in LDS
            float2 *row = P + r * PZ;
            if (c != 0) row[c] = make_float2(v[k].x, v[k].y);
            row[c + 1] = make_float2(v[k].z, v[k].w);
        }
        // pack FA + i FB: the inverse y transform then returns (X[0](y), X[NZH](y))
        if (tid >= C0T && tid < C0T + NY) {
            const float4 c0 = v[KVX - 1];
            P[(tid - C0T) * PZ] = make_float2(c0.x - c0.w, c0.y + c0.z);
        }
    }
    static_assert(NT >= NY, "one thread per row packs column 0");

    // -- forward: real plane -> main[r][c], nyq[r]: bin (ky, kz) at row pos_of<SY>(ky), column pos_of<SZH>(kz)
    LAGO_HD static void fwd_phase(int ph, int tid, const float *in, float2 *mainp, float2 *nyqp, float2 *P,
                                  float2 *tw) {
        if (ph == 0) {
            fill_twiddles(tid, tw);
            float4 v[KV];
            fwd_load(tid, in, v);
            fwd_fill(tid, v, P);
        } else if (ph <= GZ) {
            run_stage<TZ, true>(ph - 1, P, tw, tid);
        } else if (ph == GZ + 1) {
            // split the half-length transform Z into the real transform X (k = 0 .. NZH):
            // X[k] = E + w^k O, conj X[NZH-k] = E - w^k O, E = (Z[k] + conj Z[NZH-k])/2,
            // O = -i (Z[k] - conj Z[NZH-k])/2, w = exp(-2 pi i / NZ).  Z[k] sits at column pos_of(k).
            // X[0] and X[NZH] are real: they share column 0 as (X[0], X[NZH]).
            // lanes over 64 consecutive rows, k the same in every lane of a wave (RowsK): its positions, its
            // twiddle index and the two special cases cost scalar instructions
            RowsK rk(tid);
            if (rk.active) for (int k = rk.k0; k < NZH / 2 + 1; k += RowsK::SLOTS) {
                float2 *row = P + rk.y * PZ;
                if (k == 0) {
                    const float2 z = row[0];
                    row[0] = make_float2(z.x + z.y, z.x - z.y);
                } else if (k == NZH / 2) {
                    float2 *p = row + pos_of<SZH>(k);
                    *p = make_float2(p->x, -p->y);
                } else {
                    float2 *pk = row + pos_of<SZH>(k), *pm = row + pos_of<SZH>(NZH - k);
                    const float2 zk = *pk, zm = *pm;
                    const float2 E = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
                    const float2 D = make_float2(0.5f * (zk.x - zm.x), 0.5f * (zk.y + zm.y));
                    const float2 wO = cmul(make_float2(D.y, -D.x), tw[k * (LTW / NZ)]);
                    *pk = make_float2(E.x + wO.x, E.y + wO.y);
                    *pm = make_float2(E.x - wO.x, -(E.y - wO.y));
                }
            }
        } else if (ph <= GZ + 1 + GY) {
            run_stage<TY, true>(ph - GZ - 2, P, tw + TWY0, tid);
        } else if (ph == GZ + GY + 2) {
            // column 0 carried A + iB with A = X[0](y), B = X[NZH](y) both real: separate their
            // transforms FA(ky) = (P(ky) + conj P(-ky))/2, FB(ky) = -i (P(ky) - conj P(-ky))/2.
            // FA stays in column 0, FB goes to the spare column NZH.  Row ky sits at pos_of(ky).
            for (int ky = tid; ky <= NY / 2; ky += NT) {
                float2 *pk = P + pos_of<SY>(ky) * PZ, *pm = P + pos_of<SY>((NY - ky) % NY) * PZ;
                const float2 a = pk[0], b = pm[0];
                const float2 FA = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
                const float2 D = make_float2(0.5f * (a.x - b.x), 0.5f * (a.y + b.y));
                const float2 FB = make_float2(D.y, -D.x);
                pk[0] = FA;
                pk[NZH] = FB;
                pm[0] = make_float2(FA.x, -FA.y);
                pm[NZH] = make_float2(FB.x, -FB.y);
            }
        } else {
            // the spectrum goes to memory in the order it has in LDS (row r holds ky = freq_at<SY>(r), column c holds
            // kz = freq_at<SZH>(c)): no index arithmetic here nor in the inverse's fill
#pragma unroll
            for (int k = 0; k < KV; ++k) {
                if (tid + k * NT >= F4) continue;
                const int e = (tid + k * NT) * 2, r = e / NZH, c = e % NZH;
                const float2 *row = P + r * PZ;
                const float2 a = row[c], cc = row[c + 1];
                stg4<LAGO_NT_ZF_ST>(reinterpret_cast<float4 *>(mainp) + (tid + k * NT), make_float4(a.x, a.y, cc.x, cc.y));
            }
            for (int r = tid; r < NY; r += NT) nyqp[r] = P[r * PZ + NZH];
        }
    }

    // -- inverse: main[r][c], nyq[r] -> real plane (unnormalised: NY * NZ times the original)
    // oscale: applied to the finished value in the store phase -- the bits a separate multiply of the stored plane gives
    LAGO_HD static void inv_phase(int ph, int tid, float *out, const float2 *mainp, const float2 *nyqp, float2 *P,
                                  float2 *tw, float oscale = 1.0f) {
        if (ph == 0) {
            fill_twiddles(tid, tw);
            float4 v[KVX];
            inv_load(tid, mainp, nyqp, v);
            inv_fill(tid, v, P);
        } else if (ph <= GY) {
            run_stage<TY, false>(ph - 1, P, tw + TWY0, tid);
        } else if (ph == GY + 1) {
            // merge X back into the half-length transform (twice it: the missing factor 2 of the
            // real inverse): 2E = X[k] + conj X[NZH-k], 2O = (X[k] - conj X[NZH-k]) conj(w^k),
            // Z[k] = 2E + i 2O, Z[NZH-k] = conj(2E) + i conj(2O)
            RowsK rk(tid);
            if (rk.active) for (int k = rk.k0; k < NZH / 2 + 1; k += RowsK::SLOTS) {
                float2 *row = P + rk.y * PZ;
                if (k == 0) {
                    const float2 x = row[0];
                    row[0] = make_float2(x.x + x.y, x.x - x.y);
                } else if (k == NZH / 2) {
                    float2 *p = row + pos_of<SZH>(k);
                    *p = make_float2(2.0f * p->x, -2.0f * p->y);
                } else {
                    float2 *pk = row + pos_of<SZH>(k), *pm = row + pos_of<SZH>(NZH - k);
                    const float2 xk = *pk, xm = *pm;
                    const float2 E = make_float2(xk.x + xm.x, xk.y - xm.y);
                    const float2 O = cmulc(make_float2(xk.x - xm.x, xk.y + xm.y), tw[k * (LTW / NZ)]);
                    *pk = make_float2(E.x - O.y, E.y + O.x);
                    *pm = make_float2(E.x + O.y, O.x - E.y);
                }
            }
        } else if (ph <= GY + 1 + GZ) {
            run_stage<TZ, false>(ph - GY - 2, P, tw, tid);
        } else if (ph == GY + GZ + 2) {
#pragma unroll
            for (int k = 0; k < KV; ++k) {
                if (tid + k * NT >= F4) continue;
                const int e = (tid + k * NT) * 2, y = e / NZH, j = e % NZH;
                const float2 a = P[y * PZ + j], c = P[y * PZ + j + 1];
                stg4<LAGO_NT_ZI_ST>(reinterpret_cast<float4 *>(out) + (tid + k * NT),
                                    make_float4(a.x * oscale, a.y * oscale, c.x * oscale, c.y * oscale));
            }
        }
    }
    static constexpr int NPH_INV = GY + GZ + 3;
};

// ---------------------------------------------------------------------------------------------
// Planes above the LDS (256 x 256, 192 x 224, ...: ny * (nz/2 + 1) * 8 B > 160 KB): the zy pass in two launches -- ROWS
// (real transform along z of 64 rows per workgroup: the z phases of ZY on a 64-row "plane") and COLUMNS (transform along
// y of 3 components x ny x 16 kz bins per workgroup, the data flow of the x pass without an operator) -- around the same
// x pass: five launches instead of three, the same spectrum layout (row r holds ky = freq_at(r), column q holds kz =
// freq_at(q), Nyquist plane apart) and the same coefficient table.
template <class SZH_>
struct ZRows {
    static constexpr int RB = 64, NT = 512;
    using K = ZY<Sz<1, 6>, SZH_, NT>;   // only its z phases are run
    static constexpr int NZH = K::NZH, NZ = K::NZ, PZ = K::PZ, GZ = K::GZ, GYK = K::GY, KV = K::KV, F4 = K::F4;
    static constexpr size_t SMEM = K::SMEM;
    static_assert(!K::TW2, "one twiddle table");
    // forward: phases 0 .. GZ + 1 of K::fwd_phase (load, z stages, split), then this store: column 0 keeps the packed pair
    // (X[0], X[NZH]) of the row, which the column pass takes apart after ITS transform
    LAGO_HD static void fwd_store(int tid, const float2 *P, float2 *mainp) {
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            if (tid + k * NT >= F4) continue;
            const int e = (tid + k * NT) * 2, r = e / NZH, c = e % NZH;
            const float2 *row = P + r * PZ;
            const float2 a = row[c], cc = row[c + 1];
            stg4<LAGO_NT_ZF_ST>(reinterpret_cast<float4 *>(mainp) + (tid + k * NT), make_float4(a.x, a.y, cc.x, cc.y));
        }
    }
    // inverse: this fill (column 0 arrives packed from the column pass), then phases GY + 1 .. GY + GZ + 2 of K::inv_phase
    // (merge, z stages, store with the output factor)
    LAGO_HD static void inv_fill(int tid, const float2 *mainp, float2 *P) {
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            if (tid + k * NT >= F4) continue;
            const float4 v = ldg4<LAGO_NT_ZI_LD>(reinterpret_cast<const float4 *>(mainp) + (tid + k * NT));
            const int e = (tid + k * NT) * 2, r = e / NZH, c = e % NZH;
            P[r * PZ + c] = make_float2(v.x, v.y);
            P[r * PZ + c + 1] = make_float2(v.z, v.w);
        }
    }
};

struct YArgs {
    float2 *main_, *nyq;
    int nx, ny, nzh, ntile;   // ntile = ceil(nzh / 16) bin tiles per row (the last one holds 8 bins where nzh % 16 = 8)
    uint32_t total;           // nn * nx * ntile workgroups
    int rev;
};

template <class SY_, int NT_ = 256>
struct YPass {
    static constexpr int NY = SY_::N, NT = NT_, KL = 16, KCP = KL + 1;
    using T = Xf<SY_, KCP, 1, KL, 3, NY * KCP, NY, NT>;
    static constexpr int G = stage_count<SY_>();
    static constexpr int NPH = G + 3;   // forward: load | G stages | column 0 apart | store; inverse: load | column 0 packed | G stages | store
    static constexpr int ROWS_IT = NT / 8, KLD = (3 * NY + ROWS_IT - 1) / ROWS_IT;
    static constexpr bool RAGGED = 3 * NY % ROWS_IT != 0;
    static constexpr size_t SMEM = (size_t)(3 * NY * KCP + NY) * sizeof(float2);
    struct Block {   // workgroup-uniform: the three components of one (batch item, x), 16 kz bins of every row
        float2 *base, *nyq;
        size_t rs, cs, ncs;   // row stride, component stride (complex elements); component stride of the Nyquist plane
        bool first;           // the tile holds column 0 (the packed real columns kz = 0 and kz = nz/2)
        int nv;               // bins of the tile that exist (16, or nzh % 16 in a row's last tile): the other lanes load zeros, store nothing
    };
    LAGO_HD static Block locate(const YArgs &a, uint32_t blk) {
        const uint32_t qt = blk % (uint32_t)a.ntile, nxi = blk / (uint32_t)a.ntile;
        const uint32_t x = nxi % (uint32_t)a.nx, n = nxi / (uint32_t)a.nx;
        const size_t plane0 = (size_t)n * 3 * a.nx + x;
        Block b;
        b.rs = (size_t)a.nzh;
        b.cs = (size_t)a.nx * a.ny * a.nzh;
        b.ncs = (size_t)a.nx * a.ny;
        b.base = a.main_ + plane0 * a.ny * a.nzh + (size_t)qt * KL;
        b.nyq = a.nyq + plane0 * a.ny;
        b.first = qt == 0;
        b.nv = min(KL, a.nzh - (int)qt * KL);
        return b;
    }
    LAGO_HD static float2 *row_ptr(const Block &b, int rr) { return b.base + (size_t)(rr / NY) * b.cs + (size_t)(rr % NY) * b.rs; }
    LAGO_HD static void load_fill(int tid, const Block &b, float2 *buf) {
        const int rg = tid >> 3, l8 = tid & 7;
        float4 v[KLD];
#pragma unroll
        for (int k = 0; k < KLD; ++k)
            if (2 * l8 >= b.nv) v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            else if (!RAGGED || rg + k * ROWS_IT < 3 * NY) v[k] = ldg4<0>(reinterpret_cast<const float4 *>(row_ptr(b, rg + k * ROWS_IT) + 2 * l8));
#pragma unroll
        for (int k = 0; k < KLD; ++k) {
            if (RAGGED && rg + k * ROWS_IT >= 3 * NY) continue;
            float2 *d = buf + (rg + k * ROWS_IT) * KCP + 2 * l8;
            d[0] = make_float2(v[k].x, v[k].y);
            d[1] = make_float2(v[k].z, v[k].w);
        }
    }
    LAGO_HD static void store(int tid, const Block &b, const float2 *buf) {
        const int rg = tid >> 3, l8 = tid & 7;
#pragma unroll
        for (int k = 0; k < KLD; ++k) {
            if ((RAGGED && rg + k * ROWS_IT >= 3 * NY) || 2 * l8 >= b.nv) continue;
            const float2 *s = buf + (rg + k * ROWS_IT) * KCP + 2 * l8;
            const float2 a = s[0], c = s[1];
            stg4<0>(reinterpret_cast<float4 *>(row_ptr(b, rg + k * ROWS_IT) + 2 * l8), make_float4(a.x, a.y, c.x, c.y));
        }
    }
    LAGO_HD static void fill_twiddles(int tid, float2 *tw) {
        for (int t = tid; t < NY; t += NT) tw[t] = twiddle(t, NY);
    }
    // rows: natural y in, position r (ky = freq_at(r)) out
    LAGO_HD static void fwd_phase(int ph, int tid, const Block &b, float2 *buf, float2 *tw) {
        if (ph == 0) {
            fill_twiddles(tid, tw);
            load_fill(tid, b, buf);
        } else if (ph <= G) {
            run_stage<T, true>(ph - 1, buf, tw, tid);
        } else if (ph == G + 1) {
            // column 0 carried A + iB, A = X[0](y), B = X[NZH](y) both real (ZY::fwd_phase, same step): FA stays in column
            // 0, FB goes to the Nyquist plane
            if (b.first) for (int i = tid; i < 3 * (NY / 2 + 1); i += NT) {
                const int c = i / (NY / 2 + 1), ky = i % (NY / 2 + 1);
                const int rk = pos_of<SY_>(ky), rm = pos_of<SY_>((NY - ky) % NY);
                float2 *pk = buf + (c * NY + rk) * KCP, *pm = buf + (c * NY + rm) * KCP;
                const float2 a = pk[0], bb = pm[0];
                const float2 FA = make_float2(0.5f * (a.x + bb.x), 0.5f * (a.y - bb.y));
                const float2 D = make_float2(0.5f * (a.x - bb.x), 0.5f * (a.y + bb.y));
                const float2 FB = make_float2(D.y, -D.x);
                pk[0] = FA;
                pm[0] = make_float2(FA.x, -FA.y);
                b.nyq[(size_t)c * b.ncs + rk] = FB;
                b.nyq[(size_t)c * b.ncs + rm] = make_float2(FB.x, -FB.y);
            }
        } else {
            store(tid, b, buf);
        }
    }
    // rows: position r in, natural y out (unnormalised)
    LAGO_HD static void inv_phase(int ph, int tid, const Block &b, float2 *buf, float2 *tw) {
        if (ph == 0) {
            fill_twiddles(tid, tw);
            load_fill(tid, b, buf);
        } else if (ph == 1) {
            // FA + i FB into column 0: the inverse transform then returns (X[0](y), X[NZH](y)) (ZY::inv_fill, same step)
            if (b.first) for (int i = tid; i < 3 * NY; i += NT) {
                const int c = i / NY, r = i % NY;
                float2 *p = buf + (c * NY + r) * KCP;
                const float2 fa = p[0], fb = b.nyq[(size_t)c * b.ncs + r];
                p[0] = make_float2(fa.x - fb.y, fa.y + fb.x);
            }
        } else if (ph <= G + 1) {
            run_stage<T, false>(ph - 2, buf, tw, tid);
        } else {
            store(tid, b, buf);
        }
    }
};

}  // namespace fl
}  // namespace lago
