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

#include <cmath>

#include <memory>
#include <map>
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
                        int64_t nx, int64_t ny, int64_t nz, double scale, hipStream_t s, float oscale);
int fluid_xpass_launch(float *F, const float *tab, int inverse, int64_t nn, int64_t nx, int64_t ny, int64_t nzc,
                       double scale, hipStream_t s);
bool fluid2d_supported(int64_t h, int64_t w);
int fluid_metric_2d(float *out, const float *m, int inverse, const float *cosX, const float *sinX, const float *cosY,
                    const float *sinY, double alpha, double beta, double gamma, int64_t nn, int64_t h, int64_t w,
                    hipStream_t s, float oscale);
// fftg.hip
bool fluid_generic_supported(int dim, int64_t nx, int64_t ny, int64_t nz, size_t esize);
void bluestein_cache_clear();   // fftg.hip
int bluestein_cache_entries();
template <typename R>
int fluid_metric_generic(R *out, const R *m, R *work, int inverse, const R *cosX, const R *sinX, const R *cosY, const R *sinY,
                         const R *cosZ, const R *sinZ, double alpha, double beta, double gamma, int dim, int64_t nn, int64_t nx,
                         int64_t ny, int64_t nz, hipStream_t s);
// 0: rocFFT 3D plan + operator kernel; 1: rocFFT 2D (y, z) plan + fused x pass (fftx.hip);
// 2: three LDS-tiled passes, no rocFFT (fft3.hip), falling back to 1 / 0 where the shape is not supported;
// 3 (default): the same tuned passes, and the GENERIC hand-written passes (fftg.hip) for everything else -- no rocFFT.
std::atomic<int> g_fluid_xpass{3};

// Operator coefficient tables (see fftx.hip), cached like the FFT plans: one device buffer per
// (LUT generation, shape, parameters, direction), filled by a kernel on first use.  The key holds no
// pointer: a table is a function of the LUT *contents*, which the caller identifies by a generation
// number it changes whenever it passes different contents (0 = "do not cache": such calls take the
// table-free path).  Entries are shared_ptrs: a lookup keeps its table alive until its launches are
// enqueued, eviction only drops the cache's reference, and the deleter's hipFree waits for kernels
// still reading the buffer.
struct CoefTab {
    int64_t gen, nx, ny, nz;
    int inverse, device, split;
    double a, b, g;
    size_t bytes;
    float *d;
    ~CoefTab() {
        if (d) (void)hipFree(d);
    }
};
typedef std::shared_ptr<CoefTab> CoefRef;
// The cache object is heap-allocated and never destroyed: a namespace-scope vector would run ~CoefTab -> hipFree from
// a static destructor at process exit, possibly after the HIP runtime has been torn down.  Evicted and
// cleared entries are released OUTSIDE g_plan_mu: hipFree synchronises the device, and doing that under the mutex
// would stall every other thread's table lookup and hipFFT exec.
static std::vector<CoefRef> &g_tabs = *new std::vector<CoefRef>();
static const size_t kCoefCacheBytes = (size_t)1 << 30;  // at most 1 GiB of tables (24 B per frequency bin each)

struct FftPlan {
    int dim, n[3], batch, dbl, device;
    hipfftHandle fwd, inv;
    int verified;   // the plan's transforms were spot-checked against a direct DFT (below) since the last plan creation
    unsigned epoch; // value of g_plan_epoch when `verified` was last reset
};
static unsigned g_plan_epoch = 0;   // bumped by every plan creation (under g_plan_mu)
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
    // the failure this guards against depends on which other plans exist: every cached plan is checked again
    ++g_plan_epoch;
    for (FftPlan &q : g_plans) { q.verified = 0; q.epoch = g_plan_epoch; }
    p.epoch = g_plan_epoch;
    g_plans.push_back(p);
    out = p;
    return LAGO_OK;
}

// ---- spot check of a rocFFT plan -------------------------------------------------------------------------------
// rocFFT (ROCm 7.2) was caught returning a batched 2D real transform 60 % wrong once plans for other shapes existed
// (tools/probes/rocfft_2d_repro.py reproduces it with torch alone).  A third-party wrong answer must not reach the
// caller silently: the first forward and inverse execution of every cached plan is compared, for the first and the last
// transform of the batch, with a direct DFT at six frequencies / six voxels (double accumulation; a few milliseconds
// and one stream synchronisation).  A mismatch fails the call with a message naming the shape.
//  * The observed failure depends on which OTHER plans exist, so creating a plan marks every cached plan unverified
//    again (get_plan): each is re-checked on its next use.
//  * The deviation is judged against a GLOBAL scale, not against the sampled values (a centred object on a zero
//    background has corner voxels nine orders below its peak; a single sinusoid has empty bins): forward
//    |dev_k| <= tol (|X_k| + ||x||_2) -- the root-mean-square bin of a transform is ||x||_2 by Parseval and a float
//    FFT's error per bin is about eps log2(N) of that -- inverse |dev_r| <= tol (|x_r| + sqrt(sum_k w_k |X_k|^2)), the
//    root-mean-square output voxel.  tol = 1e-3 / 1e-9: three orders above the rounding noise, two to three below
//    any wrong transform.  A zero scale (m == 0: the first atlas iteration) proves nothing and leaves the plan
//    unverified.
//  * While the stream is being captured into a graph the check is skipped (it synchronises) and the plan stays
//    unverified; the scratch buffer is persistent (no hipMalloc / hipFree per check) and every value the host needs
//    arrives with ONE asynchronous copy.
constexpr int kSpot = 6;
constexpr int kSpotSlots = 2 * kSpot;          // two batch members x kSpot samples
constexpr int kSpotDoubles = kSpotSlots * 6;   // per slot: want re, want im, got re, got im, sum of squares, index

struct SpotGeom {
    int dim, n[3], nzc;          // extents (2D: n[2] = 1); bins of the last axis in the half spectrum
    long long plane, cplane;     // real / complex elements per transform
    int member[2];               // the two transforms of the batch that are checked
};

__host__ __device__ inline void spot_pick(const SpotGeom &g, int q, int (&k)[3]) {
    // frequencies / voxels: the origin, one step along each axis, two generic ones (the last axis stays inside the half spectrum)
    const int pick[kSpot][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {g.n[0] / 2, g.n[1] / 3, 0}, {3, 5, 2}};
    for (int d = 0; d < 3; ++d) k[d] = pick[q][d] % (g.n[d] > 0 ? g.n[d] : 1);
    const int last = g.dim - 1;
    if (q == 4) k[last] = g.nzc - 1;
    if (q == 3 && g.dim == 2) { k[1] = 1 % g.nzc; k[2] = 0; }
    if (k[last] >= g.nzc) k[last] = g.nzc - 1;
}

__device__ __forceinline__ double spot_block_sum(double v, double *sh) {
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// forward, after work = R2C(x): X[k] = sum_r x[r] exp(-2 pi i k.r / n) of the real input for (member, q), the bin rocFFT
// wrote there, and sum x^2 of the member
template <typename R>
__global__ __launch_bounds__(256) void spot_forward_kernel(double *out, const R *x, const R *work, SpotGeom g) {
    __shared__ double sh[256];
    const int mi = blockIdx.x / kSpot, q = blockIdx.x % kSpot;
    int k[3];
    spot_pick(g, q, k);
    const R *xm = x + (size_t)g.member[mi] * g.plane;
    double re = 0, im = 0, sq = 0;
    const long long n12 = (long long)g.n[1] * g.n[2];
    for (long long e = threadIdx.x; e < g.plane; e += 256) {
        const long long i0 = e / n12, r = e - i0 * n12, i1 = r / g.n[2], i2 = r - i1 * g.n[2];
        const double ph = -2.0 * 3.14159265358979323846 * ((double)(k[0] * i0 % g.n[0]) / g.n[0] + (double)(k[1] * i1 % g.n[1]) / g.n[1] +
                                                            (double)(k[2] * i2 % g.n[2]) / g.n[2]);
        const double v = (double)xm[e];
        re += v * cos(ph);
        im += v * sin(ph);
        sq += v * v;
    }
    re = spot_block_sum(re, sh);
    im = spot_block_sum(im, sh);
    sq = spot_block_sum(sq, sh);
    if (threadIdx.x == 0) {
        const long long idx = g.dim == 3 ? ((long long)k[0] * g.n[1] + k[1]) * g.nzc + k[2] : (long long)k[0] * g.nzc + k[1];
        const R *got = work + ((size_t)g.member[mi] * g.cplane + idx) * 2;
        double *o = out + blockIdx.x * 6;
        o[0] = re; o[1] = im; o[2] = (double)got[0]; o[3] = (double)got[1]; o[4] = sq; o[5] = (double)idx;
    }
}

// inverse (unnormalised C2R), before out = C2R(X): x[r] = sum over the half spectrum of w(k_last) Re(X[k] exp(+2 pi i k.r / n)),
// w = 1 on the two self-conjugate planes of the last axis, 2 elsewhere -- the value a correct C2R returns at voxel r --
// and sum_k w |X_k|^2 (the mean square of the output, Parseval)
template <typename R>
__global__ __launch_bounds__(256) void spot_inverse_kernel(double *out, const R *X, SpotGeom g) {
    __shared__ double sh[256];
    const int mi = blockIdx.x / kSpot, q = blockIdx.x % kSpot;
    int r3[3];
    spot_pick(g, q, r3);
    const int last = g.dim - 1;
    if (q == 4) r3[last] = g.n[last] - 1;   // (spot_pick keeps the last index inside the half spectrum: widen for voxels)
    const R *Xm = X + (size_t)g.member[mi] * g.cplane * 2;
    const int nl = g.n[last];
    const long long c12 = g.dim == 3 ? (long long)g.n[1] * g.nzc : (long long)g.nzc;
    double acc = 0, sq = 0;
    for (long long e = threadIdx.x; e < g.cplane; e += 256) {
        int kk[3] = {0, 0, 0};
        if (g.dim == 3) { kk[0] = (int)(e / c12); const long long r = e - kk[0] * c12; kk[1] = (int)(r / g.nzc); kk[2] = (int)(r - (long long)kk[1] * g.nzc); }
        else { kk[0] = (int)(e / g.nzc); kk[1] = (int)(e - (long long)kk[0] * g.nzc); }
        const int kl = kk[last];
        const double w = (kl == 0 || (nl % 2 == 0 && kl == nl / 2)) ? 1.0 : 2.0;
        double ph = 0;
        for (int d = 0; d < g.dim; ++d) ph += (double)((long long)kk[d] * r3[d] % g.n[d]) / g.n[d];
        ph *= 2.0 * 3.14159265358979323846;
        const double xr = (double)Xm[2 * e], xi = (double)Xm[2 * e + 1];
        acc += w * (xr * cos(ph) - xi * sin(ph));
        sq += w * (xr * xr + xi * xi);
    }
    acc = spot_block_sum(acc, sh);
    sq = spot_block_sum(sq, sh);
    if (threadIdx.x == 0) {
        long long idx = 0;
        for (int d = 0; d < g.dim; ++d) idx = idx * g.n[d] + r3[d];
        double *o = out + blockIdx.x * 6;
        o[0] = acc; o[1] = 0; o[4] = sq; o[5] = (double)idx;
    }
}
// ... and after it: the voxels rocFFT wrote
template <typename R>
__global__ void spot_inverse_got_kernel(double *out, const R *x, SpotGeom g) {
    const int i = threadIdx.x;
    if (i >= kSpotSlots) return;
    double *o = out + i * 6;
    o[2] = (double)x[(size_t)g.member[i / kSpot] * g.plane + (long long)o[5]];
    o[3] = 0;
}

static SpotGeom spot_geom(int dim, const int *n, int batch) {
    SpotGeom g{};
    g.dim = dim;
    for (int d = 0; d < 3; ++d) g.n[d] = d < dim ? n[d] : 1;
    g.nzc = n[dim - 1] / 2 + 1;
    g.plane = 1;
    for (int d = 0; d < dim; ++d) g.plane *= n[d];
    g.cplane = g.plane / n[dim - 1] * g.nzc;
    g.member[0] = 0;
    g.member[1] = batch - 1;
    return g;
}

static void mark_verified(const FftPlan &p) {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    for (FftPlan &q : g_plans)
        if (q.fwd == p.fwd && q.epoch == p.epoch) q.verified = 1;   // (a plan created meanwhile reset it: check again)
}

// One check at a time (first uses are rare): the scratch buffer is shared.  Heap-allocated once per device, never freed.
static std::mutex g_spot_mu;
static double *spot_scratch(int device) {
    static double *buf[64] = {};
    if (device < 0 || device >= 64) return nullptr;
    if (!buf[device] && hipMalloc((void **)&buf[device], kSpotDoubles * sizeof(double)) != hipSuccess) buf[device] = nullptr;
    return buf[device];
}
// does this execution take part in the check?  (not while the stream is captured: the check synchronises)
static bool spot_wanted(const FftPlan &p, hipStream_t s) {
    if (p.verified) return false;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st == hipStreamCaptureStatusNone;
}

// the judgement shared by both directions: 1 right, 0 nothing to judge by (zero input), -1 wrong
static int spot_judge(const double *v, bool dbl, double &worst, double &scale) {
    const double tol = dbl ? 1e-9 : 1e-3;
    int verdict = 0;
    worst = scale = 0;
    for (int i = 0; i < kSpotSlots; ++i) {
        const double *o = v + i * 6;
        const double rms = std::sqrt(o[4]);
        if (!(rms > 0)) continue;   // this member is identically zero: any transform returns zeros
        const double dr = o[2] - o[0], di = o[3] - o[1];
        const double dev = std::sqrt(dr * dr + di * di), ref = std::sqrt(o[0] * o[0] + o[1] * o[1]) + rms;
        if (!(dev <= tol * ref)) {   // (also catches NaN)
            if (verdict >= 0 || dev / ref > worst / scale) { worst = dev; scale = ref; }
            verdict = -1;
        } else if (verdict == 0) {
            verdict = 1;
        }
    }
    return verdict;
}

// Scoped first-use check of one fluid_metric call.  forward(): after `work = R2C(m)`; inverse_prepare(): before
// `out = C2R(work)`; inverse_compare(): after it.  All three are no-ops when the plan is verified or the stream is
// being captured.  verdicts: both directions right -> the plan is marked verified.
template <typename R>
struct SpotCheck {
    const FftPlan &p;
    hipStream_t s;
    bool on;
    int fwd_verdict = 0;
    SpotGeom g;
    double *d = nullptr;
    std::unique_lock<std::mutex> lk;
    SpotCheck(const FftPlan &plan, hipStream_t stream) : p(plan), s(stream), on(spot_wanted(plan, stream)) {
        if (!on) return;
        lk = std::unique_lock<std::mutex>(g_spot_mu);
        d = spot_scratch(p.device);
        if (!d) { on = false; lk.unlock(); return; }   // no scratch: the plan stays unverified
        g = spot_geom(p.dim, p.n, p.batch);
    }
    int fetch(double (&v)[kSpotDoubles]) {
        hipError_t e = hipMemcpyAsync(v, d, sizeof(v), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return fail_hip(e, "fluid_metric (rocFFT spot check)");
        return LAGO_OK;
    }
    int wrong(const char *dir, double worst, double scale) {
        return fail_invalid("fluid_metric: rocFFT returned a WRONG %s transform for %dD extents %d x %d x %d, batch %d "
                            "(spot check against a direct DFT: deviation %.3g against a scale of %.3g); see "
                            "tools/probes/rocfft_2d_repro.py", dir, p.dim, p.n[0], p.n[1], p.dim == 3 ? p.n[2] : 1, p.batch,
                            worst, scale);
    }
    int forward(const R *m, const R *work) {
        if (!on) return LAGO_OK;
        hipLaunchKernelGGL((spot_forward_kernel<R>), dim3(kSpotSlots), dim3(256), 0, s, d, m, work, g);
        double v[kSpotDoubles];
        int rc = fetch(v);
        if (rc != LAGO_OK) return rc;
        double worst, scale;
        fwd_verdict = spot_judge(v, sizeof(R) == 8, worst, scale);
        return fwd_verdict < 0 ? wrong("forward", worst, scale) : LAGO_OK;
    }
    int inverse_prepare(const R *work) {
        if (!on) return LAGO_OK;
        hipLaunchKernelGGL((spot_inverse_kernel<R>), dim3(kSpotSlots), dim3(256), 0, s, d, work, g);
        return LAGO_OK;
    }
    int inverse_compare(const R *out) {
        if (!on) return LAGO_OK;
        hipLaunchKernelGGL((spot_inverse_got_kernel<R>), dim3(1), dim3(64), 0, s, d, out, g);
        double v[kSpotDoubles];
        int rc = fetch(v);
        if (rc != LAGO_OK) return rc;
        double worst, scale;
        const int verdict = spot_judge(v, sizeof(R) == 8, worst, scale);
        if (verdict < 0) return wrong("inverse", worst, scale);
        if (verdict > 0 && fwd_verdict > 0) mark_verified(p);
        return LAGO_OK;
    }
};

static int get_coef(CoefRef &ref, int64_t gen, int inverse, const float *cosX, const float *sinX, const float *cosY,
                    const float *sinY, const float *cosZ, const float *sinZ, double alpha, double beta, double gamma,
                    int64_t nx, int64_t ny, int64_t nz, int split, hipStream_t s) {
    int device = 0;
    LAGO_HIP_TRY(hipGetDevice(&device));
    const int64_t nzc = nz / 2 + 1;
    std::vector<CoefRef> evicted;  // declared before the lock: released after it
    std::lock_guard<std::mutex> lk(g_plan_mu);
    for (const CoefRef &t : g_tabs)
        if (t->gen == gen && t->nx == nx && t->ny == ny && t->nz == nz && t->inverse == inverse && t->device == device &&
            t->split == split && t->a == alpha && t->b == beta && t->g == gamma) {
            ref = t;
            return LAGO_OK;
        }
    CoefRef t = std::make_shared<CoefTab>();
    *t = CoefTab{gen, nx, ny, nz, inverse, device, split, alpha, beta, gamma, (size_t)nx * ny * nzc * 6 * sizeof(float), nullptr};
    LAGO_HIP_TRY(hipMalloc((void **)&t->d, t->bytes));  // on failure below ~CoefTab frees it
    int rc = fluid_coef_launch(t->d, inverse, cosX, sinX, cosY, sinY, cosZ, sinZ, alpha, beta, gamma, nx, ny, nzc, split, s);
    if (rc != LAGO_OK) return rc;
    // one-time: the table is shared by later calls on ANY stream, so it must be complete before it is
    // published (steady-state calls never synchronise)
    LAGO_HIP_TRY(hipStreamSynchronize(s));
    size_t total = t->bytes;
    for (const CoefRef &o : g_tabs) total += o->bytes;
    while (!g_tabs.empty() && (total > kCoefCacheBytes || g_tabs.size() >= 64)) {  // bounded by bytes, oldest first
        total -= g_tabs.front()->bytes;
        evicted.push_back(std::move(g_tabs.front()));
        g_tabs.erase(g_tabs.begin());
    }
    g_tabs.push_back(t);
    ref = t;
    return LAGO_OK;
}

// hipFFT plans are shared process-wide and hipfftSetStream + Exec is not atomic: both are issued under the
// plan mutex (the Exec only enqueues).
// A hipFFT plan owns ONE work area: two streams executing the same plan at the same time would race on it (the equal
// sub-batches of a shoot cut over two streams, lddmm.EXPMAP_STREAMS, ask for the same (shape, batch) plan).
// Every execution therefore records an event behind itself, and an execution on ANOTHER stream first waits for the
// event of the previous one: the executions of one plan are serialised on the device, whatever streams they come from.
// (Not while a stream is being captured: no event of the outside may enter a capture; a graph's replays are ordered by
// whoever launches them.)
struct PlanUse { hipEvent_t done = nullptr; hipStream_t stream = nullptr; bool recorded = false; };
static std::map<hipfftHandle, PlanUse> *g_plan_use = nullptr;   // (under g_plan_mu; heap: never destroyed)
template <typename F>
static hipfftResult exec_on(hipfftHandle plan, hipStream_t s, F &&exec) {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    hipfftResult r = hipfftSetStream(plan, s);
    if (r != HIPFFT_SUCCESS) return r;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone;
    if (capturing) return exec();
    if (!g_plan_use) g_plan_use = new std::map<hipfftHandle, PlanUse>();
    PlanUse &u = (*g_plan_use)[plan];
    if (u.recorded && u.stream != s && hipStreamWaitEvent(s, u.done, 0) != hipSuccess) return HIPFFT_EXEC_FAILED;
    r = exec();
    if (r != HIPFFT_SUCCESS) return r;
    if (!u.done && hipEventCreateWithFlags(&u.done, hipEventDisableTiming) != hipSuccess) { u.done = nullptr; return HIPFFT_EXEC_FAILED; }
    if (hipEventRecord(u.done, s) != hipSuccess) return HIPFFT_EXEC_FAILED;
    u.stream = s;
    u.recorded = true;
    return r;
}

// float32, 3D, power-of-two nx: rocFFT does the (y, z) transforms as a batched 2D real plan, the
// x transform + operator + inverse x transform are one kernel (fftx.hip).
static int fluid_metric_xpass(float *out, const float *m, float *work, int64_t gen, int inverse, const float *cosX,
                              const float *sinX, const float *cosY, const float *sinY, const float *cosZ,
                              const float *sinZ, double alpha, double beta, double gamma, int64_t nn, int64_t nx,
                              int64_t ny, int64_t nz, hipStream_t s) {
    const int n2[3] = {(int)ny, (int)nz, 0};
    FftPlan p;
    int rc = get_plan(p, 2, n2, (int)(nn * 3 * nx), 0);
    if (rc != LAGO_OK) return rc;
    const int64_t nzc = nz / 2 + 1;
    CoefRef tab;
    rc = get_coef(tab, gen, inverse, cosX, sinX, cosY, sinY, cosZ, sinZ, alpha, beta, gamma, nx, ny, nz, 0, s);
    if (rc != LAGO_OK) return rc;
    hipfftResult r = exec_on(p.fwd, s, [&] { return hipfftExecR2C(p.fwd, (hipfftReal *)m, (hipfftComplex *)work); });
    if (r != HIPFFT_SUCCESS) return fail_fft(r, "hipfftExecR2C(2D)");
    SpotCheck<float> spot(p, s);   // an unverified plan: spot-check rocFFT against a direct DFT
    rc = spot.forward(m, work);
    if (rc != LAGO_OK) return rc;
    rc = fluid_xpass_launch(work, tab->d, inverse, nn, nx, ny, nzc, 1.0 / ((double)nx * (double)ny * (double)nz), s);
    if (rc != LAGO_OK) return rc;
    rc = spot.inverse_prepare(work);
    if (rc != LAGO_OK) return rc;
    r = exec_on(p.inv, s, [&] { return hipfftExecC2R(p.inv, (hipfftComplex *)work, (hipfftReal *)out); });
    if (r != HIPFFT_SUCCESS) return fail_fft(r, "hipfftExecC2R(2D)");
    rc = spot.inverse_compare(out);
    if (rc != LAGO_OK) return rc;
    return finish_launch(s, "fluid_metric");
}

// out *= f, in place: the trailing pass of lago_fluid_metric_scaled on the paths whose last kernel has no factor of its own
template <typename R>
__global__ __launch_bounds__(256) void scale_inplace_kernel(R *__restrict__ x, size_t n, R f) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] *= f;
}

template <typename R>
static int fluid_metric_unscaled(R *out, const R *m, R *work, int64_t gen, int inverse, const R *cosX, const R *sinX,
                                 const R *cosY, const R *sinY, const R *cosZ, const R *sinZ, double alpha, double beta,
                                 double gamma, int dim, int64_t nn, int64_t nx, int64_t ny, int64_t nz, void *stream,
                                 R oscale, bool &scaled);

// out = oscale * irfftn(operator(rfftn(m))): the factor multiplies the finished value (the bits of `out * oscale` in the
// field's precision) -- inside the last kernel of the tuned 3D passes and of the fused 2D kernel, as one more pass on
// the other paths
template <typename R>
static int fluid_metric_impl(R *out, const R *m, R *work, int64_t gen, int inverse, const R *cosX, const R *sinX, const R *cosY,
                             const R *sinY, const R *cosZ, const R *sinZ, double alpha, double beta, double gamma,
                             int dim, int64_t nn, int64_t nx, int64_t ny, int64_t nz, void *stream, double out_scale) {
    const R f = (R)out_scale;
    bool scaled = false;
    int rc = fluid_metric_unscaled<R>(out, m, work, gen, inverse, cosX, sinX, cosY, sinY, cosZ, sinZ, alpha, beta, gamma, dim,
                                      nn, nx, ny, nz, stream, f, scaled);
    if (rc != LAGO_OK || scaled || f == (R)1 || nn == 0) return rc;
    const size_t n = (size_t)nn * dim * nx * ny * (dim == 2 ? 1 : nz);
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(scale_inplace_kernel<R>, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, n, f);
    return finish_launch((hipStream_t)stream, "fluid_metric (scale)");
}

template <typename R>
static int fluid_metric_unscaled(R *out, const R *m, R *work, int64_t gen, int inverse, const R *cosX, const R *sinX,
                                 const R *cosY, const R *sinY, const R *cosZ, const R *sinZ, double alpha, double beta,
                                 double gamma, int dim, int64_t nn, int64_t nx, int64_t ny, int64_t nz, void *stream,
                                 R oscale, bool &scaled) {
    if (dim != 2 && dim != 3) return fail_invalid("Only two- and three-dimensional fluid metric is supported");
    if (dim == 2) nz = 1;
    if (nn < 0 || nx < 1 || ny < 1 || nz < 1 || nn * dim >= (1ll << 31) || nx * ny * nz >= (1ll << 29))
        return fail_invalid("fluid_metric: bad extent");
    if (nn == 0) return LAGO_OK;
    if (!out || !m || !work) return fail_invalid("fluid_metric: null pointer");
    // 2D float32 fields whose two component planes fit the LDS together: one kernel for the whole operator (fft3.hip)
    if (sizeof(R) == 4 && dim == 2 && g_fluid_xpass >= 2 && fluid2d_supported(nx, ny) && nn < (1ll << 31) &&
        (((uintptr_t)out | (uintptr_t)m) & 15) == 0) {
        note_path(LP_FLUID_2D);
        scaled = true;
        return fluid_metric_2d((float *)out, (const float *)m, inverse, (const float *)cosX, (const float *)sinX,
                               (const float *)cosY, (const float *)sinY, alpha, beta, gamma, nn, nx, ny, (hipStream_t)stream,
                               (float)oscale);
    }
    // the table-based fast paths need a LUT generation to key their cached coefficient table on
    if (gen != 0 && sizeof(R) == 4 && dim == 3 && g_fluid_xpass >= 2 && fluid_native_supported(nx, ny, nz) &&
        (((uintptr_t)out | (uintptr_t)m | (uintptr_t)work) & 15) == 0) {  // 16-byte vector accesses
        CoefRef tab;
        int rc = get_coef(tab, gen, inverse, (const float *)cosX, (const float *)sinX, (const float *)cosY,
                          (const float *)sinY, (const float *)cosZ, (const float *)sinZ, alpha, beta, gamma, nx, ny,
                          nz, 1, (hipStream_t)stream);
        if (rc != LAGO_OK) return rc;
        note_path(LP_FLUID_LDS);
        scaled = true;
        return fluid_metric_native((float *)out, (const float *)m, (float *)work, tab->d, inverse, nn, nx, ny, nz,
                                   1.0 / ((double)nx * (double)ny * (double)nz), (hipStream_t)stream, (float)oscale);
    }
    if (g_fluid_xpass >= 3 && fluid_generic_supported(dim, nx, ny, nz, sizeof(R))) {
        note_path(LP_FLUID_GENERIC);
        return fluid_metric_generic<R>(out, m, work, inverse, cosX, sinX, cosY, sinY, cosZ, sinZ, alpha, beta, gamma, dim, nn,
                                       nx, ny, nz, (hipStream_t)stream);
    }
    if (gen != 0 && sizeof(R) == 4 && dim == 3 && g_fluid_xpass && fluid_xpass_supported(nx) && nn * 3 * nx < (1ll << 31)) {
        note_path(LP_FLUID_XPASS);
        return fluid_metric_xpass((float *)out, (const float *)m, (float *)work, gen, inverse, (const float *)cosX,
                                  (const float *)sinX, (const float *)cosY, (const float *)sinY, (const float *)cosZ,
                                  (const float *)sinZ, alpha, beta, gamma, nn, nx, ny, nz, (hipStream_t)stream);
    }
    note_path(LP_FLUID_ROCFFT);
    const int n[3] = {(int)nx, (int)ny, (int)nz};
    FftPlan p;
    hipStream_t s = (hipStream_t)stream;
    int rc = get_plan(p, dim, n, (int)(nn * dim), sizeof(R) == 8);
    if (rc != LAGO_OK) return rc;
    hipfftResult r = exec_on(p.fwd, s, [&] {
        return sizeof(R) == 4 ? hipfftExecR2C(p.fwd, (hipfftReal *)m, (hipfftComplex *)work)
                              : hipfftExecD2Z(p.fwd, (hipfftDoubleReal *)m, (hipfftDoubleComplex *)work);
    });
    if (r != HIPFFT_SUCCESS) return fail_fft(r, "hipfftExec(forward)");
    SpotCheck<R> spot(p, s);   // an unverified plan: spot-check rocFFT against a direct DFT
    rc = spot.forward(m, work);
    if (rc != LAGO_OK) return rc;
    // half-spectrum extents: the last axis keeps n/2 + 1 bins
    const int64_t cx = nx, cy = dim == 2 ? ny / 2 + 1 : ny, cz = dim == 3 ? nz / 2 + 1 : 1;
    const double scale = 1.0 / ((double)nx * (double)ny * (double)nz);
    rc = fluid_operator_impl<R>(work, inverse, cosX, sinX, cosY, sinY, cosZ, sinZ, alpha, beta, gamma, dim, nn, cx, cy,
                                cz, stream, scale);
    if (rc != LAGO_OK) return rc;
    rc = spot.inverse_prepare(work);
    if (rc != LAGO_OK) return rc;
    r = exec_on(p.inv, s, [&] {
        return sizeof(R) == 4 ? hipfftExecC2R(p.inv, (hipfftComplex *)work, (hipfftReal *)out)
                              : hipfftExecZ2D(p.inv, (hipfftDoubleComplex *)work, (hipfftDoubleReal *)out);
    });
    if (r != HIPFFT_SUCCESS) return fail_fft(r, "hipfftExec(inverse)");
    rc = spot.inverse_compare(out);
    if (rc != LAGO_OK) return rc;
    return finish_launch(s, "fluid_metric");
}

}  // namespace lago

namespace lago {
void tune_generic_fuse(int on);   // fftg.hip
// mode 4: as 3 (the default), but the generic passes run the x transforms and the operator as three launches instead of the
// fused fft_xop_kernel -- the A/B and bit-comparison switch of that fusion
void tune_fluid(int mode) {
    g_fluid_xpass = mode < 0 ? 0 : (mode > 3 ? 3 : mode);
    tune_generic_fuse(mode != 4);
}
}  // namespace lago

extern "C" {
void lago_fluid_cache_clear(void) {
    lago::bluestein_cache_clear();       // the generic passes' chirp tables (fftg.hip)
    std::vector<lago::CoefRef> dropped;  // released after the lock
    std::lock_guard<std::mutex> lk(lago::g_plan_mu);
    dropped.swap(lago::g_tabs);
}
int lago_fluid_cache_entries(void) {
    std::lock_guard<std::mutex> lk(lago::g_plan_mu);
    return (int)lago::g_tabs.size();
}
int lago_fft_plan_state(int *plans, int *verified) {
    std::lock_guard<std::mutex> lk(lago::g_plan_mu);
    int nv = 0;
    for (const lago::FftPlan &p : lago::g_plans) nv += p.verified ? 1 : 0;
    if (plans) *plans = (int)lago::g_plans.size();
    if (verified) *verified = nv;
    return 0;
}
#define LAGO_DEFINE(REAL, SUF)                                                                                     \
    int lago_fluid_metric##SUF(REAL *out, const REAL *m, REAL *work, int64_t lut_generation, int inverse,         \
                               const REAL *cosX, const REAL *sinX, const REAL *cosY, const REAL *sinY,            \
                               const REAL *cosZ, const REAL *sinZ, double alpha, double beta, double gamma,       \
                               int dim, int64_t nn, int64_t nx, int64_t ny, int64_t nz, void *stream) {           \
        return lago::fluid_metric_impl<REAL>(out, m, work, lut_generation, inverse, cosX, sinX, cosY, sinY, cosZ, \
                                             sinZ, alpha, beta, gamma, dim, nn, nx, ny, nz, stream, 1.0);         \
    }                                                                                                              \
    int lago_fluid_metric_scaled##SUF(REAL *out, const REAL *m, REAL *work, int64_t lut_generation, int inverse,  \
                                      const REAL *cosX, const REAL *sinX, const REAL *cosY, const REAL *sinY,     \
                                      const REAL *cosZ, const REAL *sinZ, double alpha, double beta,              \
                                      double gamma, int dim, int64_t nn, int64_t nx, int64_t ny, int64_t nz,      \
                                      double out_scale, void *stream) {                                           \
        return lago::fluid_metric_impl<REAL>(out, m, work, lut_generation, inverse, cosX, sinX, cosY, sinY, cosZ, \
                                             sinZ, alpha, beta, gamma, dim, nn, nx, ny, nz, stream, out_scale);   \
    }
LAGO_DEFINE(float, _f32)
LAGO_DEFINE(double, _f64)
#undef LAGO_DEFINE
}
