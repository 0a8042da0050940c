// Host emulation of the fluid-metric FFT passes (lagomorph_amd/csrc/fft_lds.hpp): every phase of
// the three kernels is run for all thread ids in turn (barriers = phase boundaries) and the result
// is compared with a double-precision separable DFT + the same per-frequency operator.
// Built and run by tests/test_fft_emulation.py; needs no GPU.
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../lagomorph_amd/csrc/fft_lds.hpp"

using namespace lago::fl;
typedef std::complex<double> cd;

static double frand() { return (double)rand() / RAND_MAX * 2.0 - 1.0; }

// in-place DFT along one axis of a dense (d0, d1, d2) complex array, sign = -1 forward
static void dft_axis(std::vector<cd> &a, int d0, int d1, int d2, int axis, int sign) {
    const int dims[3] = {d0, d1, d2};
    const int n = dims[axis];
    const size_t strides[3] = {(size_t)d1 * d2, (size_t)d2, 1};
    std::vector<cd> w(n), tmp(n);
    for (int t = 0; t < n; ++t) w[t] = std::polar(1.0, sign * 2.0 * M_PI * t / n);
    const int o1 = (axis + 1) % 3, o2 = (axis + 2) % 3;
    for (int i = 0; i < dims[o1]; ++i)
        for (int j = 0; j < dims[o2]; ++j) {
            const size_t base = i * strides[o1] + j * strides[o2];
            for (int k = 0; k < n; ++k) {
                cd s = 0;
                for (int t = 0; t < n; ++t) s += a[base + t * strides[axis]] * w[(size_t)k * t % n];
                tmp[k] = s;
            }
            for (int k = 0; k < n; ++k) a[base + k * strides[axis]] = tmp[k];
        }
}

template <bool INV>
static void op_double(const float *c, cd &X, cd &Y, cd &Z) {
    const double c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c4 = c[4], c5 = c[5];
    if (INV) {
        const cd y0 = X * c0, y1 = (Y - c1 * y0) * c2, y2 = (Z - c3 * y0 - c4 * y1) * c5;
        Z = y2 * c5;
        Y = (y1 - c4 * Z) * c2;
        X = (y0 - c1 * Y - c3 * Z) * c0;
    } else {
        const cd x = c0 * X + c1 * Y + c3 * Z, y = c1 * X + c2 * Y + c4 * Z, z = c3 * X + c4 * Y + c5 * Z;
        X = x; Y = y; Z = z;
    }
}

// BIG: the zy pass as two launches (ZRows + YPass: planes above the LDS) instead of the one-plane-per-workgroup ZY kernels
template <class SX, class SY, class SZH, bool INV, int NN = 2, int XT = 256, bool BIG = false>
static int run_case() {
    constexpr int NX = SX::N, NY = SY::N, NZH = SZH::N, NZ = 2 * NZH, NZC = NZH + 1;
    using Zy = ZY<SY, SZH>;
    using Xp = XPass<SX, INV, XT>;   // XT threads per x-pass workgroup
    constexpr int ZT = Zy::THREADS;
    const size_t plane = (size_t)NY * NZ, nplanes = (size_t)NN * 3 * NX;
    std::vector<float> m(nplanes * plane), out(nplanes * plane, 0.f);
    for (auto &v : m) v = (float)frand();
    std::vector<float2> work(nplanes * NY * NZC);
    float2 *mainb = work.data(), *nyqb = work.data() + nplanes * NY * NZH;
    std::vector<float> tab((size_t)NX * NY * NZC * 6);
    for (size_t b = 0; b < (size_t)NX * NY * NZC; ++b) {
        float *c = &tab[b * 6];
        for (int e = 0; e < 6; ++e) c[e] = (float)(0.3 * frand());
        c[0] += 1.5f; c[2] += 1.5f; c[5] += 1.5f;
    }
    float *tabMw = tab.data(), *tabNw = tab.data() + (size_t)NX * NY * NZH * 6;
    // a real field needs the same operator at k and -k; inside the half spectrum that constrains
    // the planes kz = 0 and kz = NZH (the fluid operator's LUTs satisfy it by construction)
    for (int kx = 0; kx < NX; ++kx)
        for (int ky = 0; ky < NY; ++ky) {
            const size_t a = (size_t)kx * NY + ky, bb = (size_t)((NX - kx) % NX) * NY + (NY - ky) % NY;
            for (int e = 0; e < 6; ++e) {
                tabMw[bb * NZH * 6 + e] = tabMw[a * NZH * 6 + e];
                tabNw[bb * 6 + e] = tabNw[a * 6 + e];
            }
        }
    // the passes keep the spectrum in the order the transforms leave in LDS (row r = pos_of(ky), column q = pos_of(kz));
    // the x pass takes its coefficient table in the same order
    std::vector<float> tabP(tab.size());
    for (int kx = 0; kx < NX; ++kx)
        for (int ky = 0; ky < NY; ++ky) {
            const size_t src = (size_t)kx * NY + ky, dst = (size_t)kx * NY + pos_of<SY>(ky);
            for (int kz = 0; kz < NZH; ++kz)
                for (int e = 0; e < 6; ++e) tabP[(dst * NZH + pos_of<SZH>(kz)) * 6 + e] = tabMw[(src * NZH + kz) * 6 + e];
            for (int e = 0; e < 6; ++e) tabP[((size_t)NX * NY * NZH + dst) * 6 + e] = tabNw[src * 6 + e];
        }
    const float *tabM = tabMw, *tabN = tabNw;
    const float scale = 1.0f / ((float)NX * NY * NZ);

    using ZR = ZRows<SZH>;
    using ZK = typename ZR::K;
    using Yp = YPass<SY, 256>;
    std::vector<float2> lds(Zy::SMEM / sizeof(float2) + Xp::SMEM / sizeof(float2) + ZR::SMEM / sizeof(float2) + Yp::SMEM / sizeof(float2));
    YArgs ya;
    ya.main_ = mainb; ya.nyq = nyqb; ya.nx = NX; ya.ny = NY; ya.nzh = NZH; ya.ntile = (NZH + 15) / 16;
    ya.total = (uint32_t)(NN * NX * ya.ntile); ya.rev = 0;
    const size_t nblocks = nplanes * NY / ZR::RB;
    // zy forward
    if (BIG) {
        for (size_t blk = 0; blk < nblocks; ++blk) {
            float2 *P = lds.data(), *tw = P + ZR::RB * ZR::PZ;
            for (int ph = 0; ph <= ZR::GZ + 1; ++ph)
                for (int tid = 0; tid < ZR::NT; ++tid)
                    ZK::fwd_phase(ph, tid, m.data() + blk * ZR::RB * NZ, nullptr, nullptr, P, tw);
            for (int tid = 0; tid < ZR::NT; ++tid) ZR::fwd_store(tid, P, mainb + blk * ZR::RB * NZH);
        }
        for (uint32_t blk = 0; blk < ya.total; ++blk) {
            const auto b = Yp::locate(ya, blk);
            float2 *buf = lds.data(), *tw = buf + 3 * NY * Yp::KCP;
            for (int ph = 0; ph < Yp::NPH; ++ph)
                for (int tid = 0; tid < 256; ++tid) Yp::fwd_phase(ph, tid, b, buf, tw);
        }
    } else
    for (size_t p = 0; p < nplanes; ++p) {
        float2 *P = lds.data(), *tw = P + NY * Zy::PZ;
        for (int ph = 0; ph < Zy::NPH; ++ph)
            for (int tid = 0; tid < ZT; ++tid)
                Zy::fwd_phase(ph, tid, m.data() + p * plane, mainb + p * NY * NZH, nyqb + p * NY, P, tw);
    }
    // check the spectrum of plane-wise 2D transforms against the DFT (first batch item, component 1)
    double err2d = 0, ref2d = 0;
    {
        std::vector<cd> a((size_t)NX * NY * NZ);
        for (size_t i = 0; i < a.size(); ++i) a[i] = m[(size_t)1 * NX * plane + i];
        dft_axis(a, NX, NY, NZ, 2, -1);
        dft_axis(a, NX, NY, NZ, 1, -1);
        for (int x = 0; x < NX; ++x)
            for (int ky = 0; ky < NY; ++ky)
                for (int kz = 0; kz <= NZH; ++kz) {
                    const size_t pl = (size_t)1 * NX + x;
                    const size_t rr = pos_of<SY>(ky);
                    const float2 g = kz < NZH ? mainb[(pl * NY + rr) * NZH + pos_of<SZH>(kz)] : nyqb[pl * NY + rr];
                    const cd r = a[((size_t)x * NY + ky) * NZ + kz];
                    err2d = std::max(err2d, std::abs(cd(g.x, g.y) - r));
                    ref2d = std::max(ref2d, std::abs(r));
                }
    }
    // x pass
    XArgs xa;
    xa.main_ = mainb; xa.nyq = nyqb; xa.tabM = tabP.data(); xa.tabN = tabP.data() + (size_t)NX * NY * NZH * 6;
    xa.ny = NY; xa.nzh = NZH; xa.nch = NY * NZH / 16; xa.items_per_n = xa.nch + (NY + 15) / 16;
    xa.nn = NN; xa.ipw = 1; xa.scale = scale; xa.total = (uint32_t)(NN * xa.items_per_n);
    std::vector<typename Xp::Regs> regs(XT);
    for (uint32_t blk = 0; blk < xa.total; ++blk) {
        const auto b = Xp::locate(xa, blk);
        float2 *buf = lds.data(), *tw = buf + 3 * NX * Xp::KCP;
        for (int ph = 0; ph < Xp::NPH; ++ph)
            for (int tid = 0; tid < XT; ++tid) Xp::phase(ph, tid, regs[tid], b, buf, tw, xa.scale);
    }
    // zy inverse
    if (BIG) {
        for (uint32_t blk = 0; blk < ya.total; ++blk) {
            const auto b = Yp::locate(ya, blk);
            float2 *buf = lds.data(), *tw = buf + 3 * NY * Yp::KCP;
            for (int ph = 0; ph < Yp::NPH; ++ph)
                for (int tid = 0; tid < 256; ++tid) Yp::inv_phase(ph, tid, b, buf, tw);
        }
        for (size_t blk = 0; blk < nblocks; ++blk) {
            float2 *P = lds.data(), *tw = P + ZR::RB * ZR::PZ;
            for (int tid = 0; tid < ZR::NT; ++tid) ZK::fill_twiddles(tid, tw);
            for (int tid = 0; tid < ZR::NT; ++tid) ZR::inv_fill(tid, mainb + blk * ZR::RB * NZH, P);
            for (int ph = ZR::GYK + 1; ph <= ZR::GYK + ZR::GZ + 2; ++ph)
                for (int tid = 0; tid < ZR::NT; ++tid)
                    ZK::inv_phase(ph, tid, out.data() + blk * ZR::RB * NZ, nullptr, nullptr, P, tw);
        }
    } else
    for (size_t p = 0; p < nplanes; ++p) {
        float2 *P = lds.data(), *tw = P + NY * Zy::PZ;
        for (int ph = 0; ph < Zy::NPH_INV; ++ph)
            for (int tid = 0; tid < ZT; ++tid)
                Zy::inv_phase(ph, tid, out.data() + p * plane, mainb + p * NY * NZH, nyqb + p * NY, P, tw);
    }
    // reference: full complex 3D DFT in double, operator on every bin, inverse DFT
    double err = 0, ref = 0;
    for (int n = 0; n < NN; ++n) {
        std::vector<cd> a[3];
        for (int c = 0; c < 3; ++c) {
            a[c].resize((size_t)NX * NY * NZ);
            for (size_t i = 0; i < a[c].size(); ++i) a[c][i] = m[((size_t)n * 3 + c) * NX * plane + i];
            for (int ax = 2; ax >= 0; --ax) dft_axis(a[c], NX, NY, NZ, ax, -1);
        }
        for (int kx = 0; kx < NX; ++kx)
            for (int ky = 0; ky < NY; ++ky)
                for (int kz = 0; kz < NZ; ++kz) {
                    // the operator of bin (kx, ky, kz > NZH) is the conjugate-symmetric partner's: the
                    // table covers the half spectrum, and the tabulated coefficients are real
                    const int hx = kz <= NZH ? kx : (NX - kx) % NX, hy = kz <= NZH ? ky : (NY - ky) % NY,
                              hz = kz <= NZH ? kz : NZ - kz;
                    const float *c = hz < NZH ? tabM + (((size_t)hx * NY + hy) * NZH + hz) * 6
                                              : tabN + ((size_t)hx * NY + hy) * 6;
                    const size_t i = ((size_t)kx * NY + ky) * NZ + kz;
                    op_double<INV>(c, a[0][i], a[1][i], a[2][i]);
                }
        for (int c = 0; c < 3; ++c) {
            for (int ax = 0; ax < 3; ++ax) dft_axis(a[c], NX, NY, NZ, ax, +1);
            for (size_t i = 0; i < a[c].size(); ++i) {
                const double r = a[c][i].real() / ((double)NX * NY * NZ);
                err = std::max(err, std::fabs(r - (double)out[((size_t)n * 3 + c) * NX * plane + i]));
                ref = std::max(ref, std::fabs(r));
            }
        }
    }
    const bool ok = err2d <= 2e-5 * ref2d && err <= 2e-5 * ref;
    printf("%s nx=%d ny=%d nz=%d inverse=%d: 2D spectrum err %.3e (scale %.3e), result err %.3e (scale %.3e)\n",
           ok ? "ok  " : "FAIL", NX, NY, NZ, (int)INV, err2d, ref2d, err, ref);
    return ok ? 0 : 1;
}

int main() {
    int bad = 0;
    // powers of two
    bad += run_case<Sz<1, 6>, Sz<1, 5>, Sz<1, 5>, false>();   // 64 x 32 x 64
    bad += run_case<Sz<1, 6>, Sz<1, 6>, Sz<1, 6>, true>();    // 64 x 64 x 128
    bad += run_case<Sz<1, 7>, Sz<1, 5>, Sz<1, 5>, true>();    // 128 x 32 x 64
    // radix 3 and 5 times a power of two (96 = 3*32, 160 = 5*32; half lengths 48 = 3*16, 80 = 5*16)
    bad += run_case<Sz<3, 5>, Sz<1, 5>, Sz<3, 4>, false>();   // 96 x 32 x 96
    bad += run_case<Sz<5, 5>, Sz<1, 5>, Sz<5, 4>, true>();    // 160 x 32 x 160
    bad += run_case<Sz<1, 6>, Sz<5, 5>, Sz<1, 5>, false>();   // 64 x 160 x 64
    bad += run_case<Sz<1, 6>, Sz<3, 5>, Sz<5, 4>, true>();    // 64 x 96 x 160
    // four-level groups (256 = two groups of four, 128 = 4 + 3) and the unfused radix-3 level in front of 3 + 3 levels
    bad += run_case<Sz<1, 8>, Sz<1, 5>, Sz<1, 5>, false>();   // 256 x 32 x 64
    bad += run_case<Sz<1, 5>, Sz<3, 6>, Sz<1, 7>, true>();    // 32 x 192 x 256
    bad += run_case<Sz<1, 5>, Sz<1, 7>, Sz<3, 5>, false>();   // 32 x 128 x 192
    // radix 11 and 13 (176 = 11 * 16, 208 = 13 * 16; half lengths 88 = 11 * 8, 104 = 13 * 8): ragged x-pass rows, tiles of
    // 16 positions that straddle the 88-bin rows, a twiddle table per axis where the common one does not fit (208 x 176)
    bad += run_case<Sz<11, 4>, Sz<1, 5>, Sz<1, 5>, true>();    // 176 x 32 x 64
    bad += run_case<Sz<13, 4>, Sz<1, 5>, Sz<1, 5>, false, 2, 512>();   // 208 x 32 x 64, the 512-thread x pass (6 1/2 operator rows per thread)
    bad += run_case<Sz<1, 4>, Sz<13, 4>, Sz<11, 3>, true, 1>();   // 16 x 208 x 176 (one batch item)
    bad += run_case<Sz<1, 4>, Sz<11, 4>, Sz<13, 3>, false, 1>();  // 16 x 176 x 208 (one batch item)
    // odd factors 7, 9 (not a prime: the pair form of the odd level holds for any odd radix) and 15: 112 = 7 * 16, 144 = 9 * 16,
    // 240 = 15 * 16 (512-thread x pass, 7 1/2 operator rows per thread), half lengths 56 = 7 * 8 and 72 = 9 * 8
    bad += run_case<Sz<7, 4>, Sz<1, 5>, Sz<1, 5>, true>();           // 112 x 32 x 64
    bad += run_case<Sz<1, 4>, Sz<7, 4>, Sz<7, 3>, false, 1>();       // 16 x 112 x 112
    bad += run_case<Sz<1, 4>, Sz<9, 4>, Sz<9, 3>, true, 1>();        // 16 x 144 x 144
    bad += run_case<Sz<15, 4>, Sz<1, 5>, Sz<1, 5>, false, 1, 512>(); // 240 x 32 x 64
    // ny % 16 = 8: the last tile of the Nyquist plane holds 8 bins (the other lanes load zeros and store nothing);
    // half lengths 44 = 11 * 4 and 52 = 13 * 4 (two radix-2 levels), 88-point x lines (5 1/2 operator rows per thread)
    bad += run_case<Sz<11, 3>, Sz<13, 3>, Sz<11, 2>, true, 1>();     // 88 x 104 x 88
    bad += run_case<Sz<13, 3>, Sz<11, 3>, Sz<13, 2>, false, 1>();    // 104 x 88 x 104 (x lengths with NX % 16 = 0 do not take such planes)
    // planes above the LDS: rows + columns around the x pass (ZRows, YPass) -- on small shapes here, the code is the same
    bad += run_case<Sz<1, 4>, Sz<1, 6>, Sz<1, 5>, true, 2, 256, true>();     // 16 x 64 x 64
    bad += run_case<Sz<1, 4>, Sz<3, 6>, Sz<3, 4>, false, 1, 256, true>();    // 16 x 192 x 96
    bad += run_case<Sz<1, 5>, Sz<7, 4>, Sz<1, 6>, true, 1, 256, true>();     // 32 x 112 x 128
    bad += run_case<Sz<1, 4>, Sz<1, 6>, Sz<11, 3>, false, 1, 256, true>();   // 16 x 64 x 176 (88 bins per row: the column pass's half tile)
    printf(bad ? "FAILED\n" : "all ok\n");
    return bad;
}
