// FluidMetric sharp/flat as one C-ABI call: rFFT -> per-frequency operator -> inverse rFFT.
//
// Replaces FluidMetricOperator.forward of the reference
// (/root/reference/lagomorph/metric.py:11-19: torch.rfft + lagomorph_ext.fluid_operator +
// torch.irfft).  The transforms are rocFFT through hipFFT, called directly on caller-provided
// buffers: no defensive copies of the real input or of the half-spectrum (torch.fft makes three
// per call on ROCm), and the two 1/sqrt(N) "ortho" scalings collapse into one 1/N factor applied
// inside the per-frequency kernel.  The operator is linear, so this equals the reference's
// ortho -> operator -> ortho pipeline up to rounding.
//
// hipFFT plans are cached per (rank, extents, batch, precision); creating a plan allocates rocFFT
// work memory once, the calls themselves allocate nothing.
#include <hipfft/hipfft.h>

#include <string.h>

#include <mutex>
#include <vector>

#include "common.hpp"

namespace lago {

template <typename R>
int fluid_operator_impl(R *Fm, int inverse, const R *cosX, const R *sinX, const R *cosY, const R *sinY,
                        const R *cosZ, const R *sinZ, double alpha, double beta, double gamma, int dim,
                        int64_t nn, int64_t nx, int64_t ny, int64_t nz, void *stream, double scale);  // metric.hip

// fftx.hip
int fluid_coef_launch(float *tab, int inverse, const float *cosX, const float *sinX, const float *cosY,
                      const float *sinY, const float *cosZ, const float *sinZ, double alpha, double beta,
                      double gamma, int64_t nx, int64_t ny, int64_t nzc, int split, hipStream_t s);
bool fluid_xpass_supported(int64_t nx);
// fft3.hip
bool fluid_native_supported(int64_t nx, int64_t ny, int64_t nz);
int fluid_metric_native(float *out, const float *m, float *work, const float *tab, int inverse, int64_t nn,
                        int64_t nx, int64_t ny, int64_t nz, double scale, hipStream_t s);
int fluid_xpass_launch(float *F, const float *tab, int inverse, int64_t nn, int64_t nx, int64_t ny, int64_t nzc,
                       double scale, hipStream_t s);
// 0: rocFFT 3D plan + operator kernel; 1: rocFFT 2D (y, z) plan + fused x pass (fftx.hip);
// 2: three LDS-tiled passes, no rocFFT (fft3.hip).  Each falls back to the previous one where the
// shape is not supported.
int g_fluid_xpass = 2;

// Operator coefficient tables (see fftx.hip), cached like the FFT plans: one device buffer per
// (shape, parameters, direction, LUT identity), filled by a kernel on first use.
struct CoefTab {
    int64_t nx, ny, nzc;
    int inverse, device, split;
    double a, b, g;
    const void *luts[6];
    float *d;
};
static std::vector<CoefTab> g_tabs;

struct FftPlan {
    int dim, n[3], batch, dbl, device;
    hipfftHandle fwd, inv;
};
static std::vector<FftPlan> g_plans;
static std::mutex g_plan_mu;

static int fail_fft(hipfftResult r, const char *what) { return fail_invalid("hipFFT error %d in %s", (int)r, what); }

static int get_plan(FftPlan &out, int dim, const int *n, int batch, int dbl) {
    int device = 0;
    LAGO_HIP_TRY(hipGetDevice(&device));
    std::lock_guard<std::mutex> lk(g_plan_mu);
    for (const FftPlan &p : g_plans)
        if (p.dim == dim && p.batch == batch && p.dbl == dbl && p.device == device && p.n[0] == n[0] &&
            p.n[1] == n[1] && (dim == 2 || p.n[2] == n[2])) {
            out = p;
            return LAGO_OK;
        }
    FftPlan p{};
    p.dim = dim;
    p.batch = batch;
    p.dbl = dbl;
    p.device = device;
    for (int d = 0; d < dim; ++d) p.n[d] = n[d];
    int nn[3] = {n[0], n[1], dim == 3 ? n[2] : 0};
    hipfftResult r = hipfftPlanMany(&p.fwd, dim, nn, nullptr, 1, 0, nullptr, 1, 0, dbl ? HIPFFT_D2Z : HIPFFT_R2C, batch);
    if (r != HIPFFT_SUCCESS) return fail_fft(r, "hipfftPlanMany(R2C)");
    r = hipfftPlanMany(&p.inv, dim, nn, nullptr, 1, 0, nullptr, 1, 0, dbl ? HIPFFT_Z2D : HIPFFT_C2R, batch);
    if (r != HIPFFT_SUCCESS) return fail_fft(r, "hipfftPlanMany(C2R)");
    g_plans.push_back(p);
    out = p;
    return LAGO_OK;
}

static int get_coef(float *&tab, int inverse, const float *cosX, const float *sinX, const float *cosY,
                    const float *sinY, const float *cosZ, const float *sinZ, double alpha, double beta, double gamma,
                    int64_t nx, int64_t ny, int64_t nzc, int split, hipStream_t s) {
    int device = 0;
    LAGO_HIP_TRY(hipGetDevice(&device));
    const void *l[6] = {cosX, sinX, cosY, sinY, cosZ, sinZ};
    std::lock_guard<std::mutex> lk(g_plan_mu);
    for (const CoefTab &t : g_tabs)
        if (t.nx == nx && t.ny == ny && t.nzc == nzc && t.inverse == inverse && t.device == device && t.split == split &&
            t.a == alpha &&
            t.b == beta && t.g == gamma && !memcmp(t.luts, l, sizeof(l))) {
            tab = t.d;
            return LAGO_OK;
        }
    CoefTab t{nx, ny, nzc, inverse, device, split, alpha, beta, gamma, {cosX, sinX, cosY, sinY, cosZ, sinZ}, nullptr};
    LAGO_HIP_TRY(hipMalloc((void **)&t.d, (size_t)nx * ny * nzc * 6 * sizeof(float)));
    int rc = fluid_coef_launch(t.d, inverse, cosX, sinX, cosY, sinY, cosZ, sinZ, alpha, beta, gamma, nx, ny, nzc, split, s);
    if (rc != LAGO_OK) return rc;
    // one-time: the table is shared by later calls on ANY stream, so it must be complete before it is
    // published (steady-state calls never synchronise)
    LAGO_HIP_TRY(hipStreamSynchronize(s));
    if (g_tabs.size() >= 16) {  // bounded cache
        (void)hipFree(g_tabs.front().d);
        g_tabs.erase(g_tabs.begin());
    }
    g_tabs.push_back(t);
    tab = t.d;
    return LAGO_OK;
}

// float32, 3D, power-of-two nx: rocFFT does the (y, z) transforms as a batched 2D real plan, the
// x transform + operator + inverse x transform are one kernel (fftx.hip).
static int fluid_metric_xpass(float *out, const float *m, float *work, int inverse, const float *cosX,
                              const float *sinX, const float *cosY, const float *sinY, const float *cosZ,
                              const float *sinZ, double alpha, double beta, double gamma, int64_t nn, int64_t nx,
                              int64_t ny, int64_t nz, hipStream_t s) {
    const int n2[3] = {(int)ny, (int)nz, 0};
    FftPlan p;
    int rc = get_plan(p, 2, n2, (int)(nn * 3 * nx), 0);
    if (rc != LAGO_OK) return rc;
    const int64_t nzc = nz / 2 + 1;
    float *tab = nullptr;
    rc = get_coef(tab, inverse, cosX, sinX, cosY, sinY, cosZ, sinZ, alpha, beta, gamma, nx, ny, nzc, 0, s);
    if (rc != LAGO_OK) return rc;
    hipfftResult r = hipfftSetStream(p.fwd, s);
    if (r != HIPFFT_SUCCESS) return fail_fft(r, "hipfftSetStream");
    r = hipfftSetStream(p.inv, s);
    if (r != HIPFFT_SUCCESS) return fail_fft(r, "hipfftSetStream");
    r = hipfftExecR2C(p.fwd, (hipfftReal *)m, (hipfftComplex *)work);
    if (r != HIPFFT_SUCCESS) return fail_fft(r, "hipfftExecR2C(2D)");
    rc = fluid_xpass_launch(work, tab, inverse, nn, nx, ny, nzc, 1.0 / ((double)nx * (double)ny * (double)nz), s);
    if (rc != LAGO_OK) return rc;
    r = hipfftExecC2R(p.inv, (hipfftComplex *)work, (hipfftReal *)out);
    if (r != HIPFFT_SUCCESS) return fail_fft(r, "hipfftExecC2R(2D)");
    return finish_launch(s, "fluid_metric");
}

template <typename R>
static int fluid_metric_impl(R *out, const R *m, R *work, int inverse, const R *cosX, const R *sinX, const R *cosY,
                             const R *sinY, const R *cosZ, const R *sinZ, double alpha, double beta, double gamma,
                             int dim, int64_t nn, int64_t nx, int64_t ny, int64_t nz, void *stream) {
    if (dim != 2 && dim != 3) return fail_invalid("Only two- and three-dimensional fluid metric is supported");
    if (dim == 2) nz = 1;
    if (nn < 0 || nx < 1 || ny < 1 || nz < 1 || nn * dim >= (1ll << 31) || nx * ny * nz >= (1ll << 29))
        return fail_invalid("fluid_metric: bad extent");
    if (nn == 0) return LAGO_OK;
    if (!out || !m || !work) return fail_invalid("fluid_metric: null pointer");
    if (sizeof(R) == 4 && dim == 3 && g_fluid_xpass >= 2 && fluid_native_supported(nx, ny, nz) &&
        (((uintptr_t)out | (uintptr_t)m | (uintptr_t)work) & 15) == 0) {  // 16-byte vector accesses
        float *tab = nullptr;
        int rc = get_coef(tab, inverse, (const float *)cosX, (const float *)sinX, (const float *)cosY,
                          (const float *)sinY, (const float *)cosZ, (const float *)sinZ, alpha, beta, gamma, nx, ny,
                          nz / 2 + 1, 1, (hipStream_t)stream);
        if (rc != LAGO_OK) return rc;
        return fluid_metric_native((float *)out, (const float *)m, (float *)work, tab, inverse, nn, nx, ny, nz,
                                   1.0 / ((double)nx * (double)ny * (double)nz), (hipStream_t)stream);
    }
    if (sizeof(R) == 4 && dim == 3 && g_fluid_xpass && fluid_xpass_supported(nx) && nn * 3 * nx < (1ll << 31))
        return fluid_metric_xpass((float *)out, (const float *)m, (float *)work, inverse, (const float *)cosX,
                                  (const float *)sinX, (const float *)cosY, (const float *)sinY, (const float *)cosZ,
                                  (const float *)sinZ, alpha, beta, gamma, nn, nx, ny, nz, (hipStream_t)stream);
    const int n[3] = {(int)nx, (int)ny, (int)nz};
    FftPlan p;
    int rc = get_plan(p, dim, n, (int)(nn * dim), sizeof(R) == 8);
    if (rc != LAGO_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipfftResult r = hipfftSetStream(p.fwd, s);
    if (r != HIPFFT_SUCCESS) return fail_fft(r, "hipfftSetStream");
    r = hipfftSetStream(p.inv, s);
    if (r != HIPFFT_SUCCESS) return fail_fft(r, "hipfftSetStream");
    if (sizeof(R) == 4)
        r = hipfftExecR2C(p.fwd, (hipfftReal *)m, (hipfftComplex *)work);
    else
        r = hipfftExecD2Z(p.fwd, (hipfftDoubleReal *)m, (hipfftDoubleComplex *)work);
    if (r != HIPFFT_SUCCESS) return fail_fft(r, "hipfftExec(forward)");
    // half-spectrum extents: the last axis keeps n/2 + 1 bins
    const int64_t cx = nx, cy = dim == 2 ? ny / 2 + 1 : ny, cz = dim == 3 ? nz / 2 + 1 : 1;
    const double scale = 1.0 / ((double)nx * (double)ny * (double)nz);
    rc = fluid_operator_impl<R>(work, inverse, cosX, sinX, cosY, sinY, cosZ, sinZ, alpha, beta, gamma, dim, nn, cx, cy,
                                cz, stream, scale);
    if (rc != LAGO_OK) return rc;
    if (sizeof(R) == 4)
        r = hipfftExecC2R(p.inv, (hipfftComplex *)work, (hipfftReal *)out);
    else
        r = hipfftExecZ2D(p.inv, (hipfftDoubleComplex *)work, (hipfftDoubleReal *)out);
    if (r != HIPFFT_SUCCESS) return fail_fft(r, "hipfftExec(inverse)");
    return finish_launch(s, "fluid_metric");
}

}  // namespace lago

extern "C" {
void lago_set_fluid_xpass(int mode) { lago::g_fluid_xpass = mode < 0 ? 0 : (mode > 2 ? 2 : mode); }
#define LAGO_DEFINE(REAL, SUF)                                                                                     \
    int lago_fluid_metric##SUF(REAL *out, const REAL *m, REAL *work, int inverse, const REAL *cosX,               \
                               const REAL *sinX, const REAL *cosY, const REAL *sinY, const REAL *cosZ,            \
                               const REAL *sinZ, double alpha, double beta, double gamma, int dim, int64_t nn,    \
                               int64_t nx, int64_t ny, int64_t nz, void *stream) {                                \
        return lago::fluid_metric_impl<REAL>(out, m, work, inverse, cosX, sinX, cosY, sinY, cosZ, sinZ, alpha,    \
                                             beta, gamma, dim, nn, nx, ny, nz, stream);                           \
    }
LAGO_DEFINE(float, _f32)
LAGO_DEFINE(double, _f64)
#undef LAGO_DEFINE
}
