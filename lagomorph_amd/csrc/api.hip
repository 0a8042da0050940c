// Housekeeping half of the C ABI: debug flag, error reporting, version.
#include <stdarg.h>
#include <string.h>

#include "common.hpp"

namespace lago {

static std::atomic<int> g_debug{0};
std::atomic<int> g_splat_mode{1};
std::atomic<int> g_interp_vec{1};
std::atomic<long long> g_path_launches[LP_COUNT];
std::atomic<int> g_launch_alt{1};
std::atomic<unsigned> g_launch_seq{0};
std::atomic<long long> g_reversed_launches{0};
static thread_local char g_err[512] = "";

int fail_invalid(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return LAGO_ERR_INVALID;
}

int fail_hip(hipError_t e, const char *what) {
    snprintf(g_err, sizeof(g_err), "HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return LAGO_ERR_HIP;
}

// The reference's LAGOMORPH_CUDA_CHECK (include/defs.h:17-23) synchronises and
// prints in debug mode; here the fault is returned to the caller instead.
int finish_launch(hipStream_t s, const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, what);
    if (g_debug) {
        // (not while the stream is being captured into a graph: a synchronisation would invalidate the capture)
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone) return LAGO_OK;
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) return fail_hip(e, what);
    }
    return LAGO_OK;
}

// ---- tuning: one struct (include/lagomorph_hip.h).  The struct in force is kept here; a set applies it to the modules.
static std::mutex g_tuning_mu;
static lago_tuning default_tuning() {
    lago_tuning t;
    memset(&t, 0, sizeof(t));
    t.struct_size = (uint32_t)sizeof(lago_tuning);
    t.splat_mode = 1;
    const int32_t tile[7] = {0, 8, 0, 1, 1, 4, 512}, shear[8] = {1, 8, 6, 0, 1, 1, 4, 1024};
    memcpy(t.splat_tile, tile, sizeof(tile));
    memcpy(t.splat_shear, shear, sizeof(shear));
    t.splat_shear_mc = 2;
    t.splat_mc = 1;
    t.vector_kernels = 1;
    t.launch_order = 1;
    t.stencil_tile = 1;
    t.gather_window = 1;
    t.fluid_mode = 3;
    t.fluid_xpass_ipw = 0;
    t.fluid_zy_persist = 1;
    t.fluid_xpass_wide = 1;
    t.fluid_xpass_persist = 1;
    t.affine_box = 1;
    return t;
}
static lago_tuning g_tuning = default_tuning();

}  // namespace lago

extern "C" {
void lago_default_tuning(lago_tuning *t) {
    if (!t) return;
    const lago_tuning d = lago::default_tuning();
    const uint32_t n = t->struct_size < sizeof(d) ? t->struct_size : (uint32_t)sizeof(d);
    if (n > sizeof(uint32_t)) memcpy((char *)t + sizeof(uint32_t), (const char *)&d + sizeof(uint32_t), n - sizeof(uint32_t));
}
void lago_get_tuning(lago_tuning *t) {
    if (!t) return;
    std::lock_guard<std::mutex> lk(lago::g_tuning_mu);
    const uint32_t n = t->struct_size < sizeof(lago::g_tuning) ? t->struct_size : (uint32_t)sizeof(lago::g_tuning);
    if (n > sizeof(uint32_t))
        memcpy((char *)t + sizeof(uint32_t), (const char *)&lago::g_tuning + sizeof(uint32_t), n - sizeof(uint32_t));
}
int lago_set_tuning(const lago_tuning *t) {
    if (!t) return lago::fail_invalid("lago_set_tuning: null pointer");
    if (t->struct_size < sizeof(uint32_t) || t->struct_size % sizeof(int32_t) != 0)
        return lago::fail_invalid("lago_set_tuning: struct_size %u is not a whole number of fields", t->struct_size);
    std::lock_guard<std::mutex> lk(lago::g_tuning_mu);
    lago_tuning &g = lago::g_tuning;
    const uint32_t n = t->struct_size < sizeof(g) ? t->struct_size : (uint32_t)sizeof(g);
    memcpy((char *)&g + sizeof(uint32_t), (const char *)t + sizeof(uint32_t), n - sizeof(uint32_t));
    lago::g_splat_mode = g.splat_mode;
    lago::g_interp_vec = g.vector_kernels ? 1 : 0;
    lago::g_launch_alt = g.launch_order ? 1 : 0;
    lago::tune_splat(g.splat_tile, g.splat_shear, g.splat_shear_mc, g.splat_mc);
    lago::tune_fused(g.stencil_tile, g.gather_window);
    lago::tune_fluid(g.fluid_mode);
    lago::tune_fluid_passes(g.fluid_xpass_ipw, g.fluid_zy_persist, g.fluid_xpass_wide, g.fluid_xpass_persist);
    lago::tune_affine(g.affine_box);
    return LAGO_OK;
}
void lago_set_debug(int on) { lago::g_debug = on ? 1 : 0; }
int lago_get_debug(void) { return lago::g_debug; }
int lago_abi_version(void) { return LAGO_ABI_VERSION; }
const char *lago_version(void) { return "lagomorph_hip 0.1 (gfx950, HIP)"; }
const char *lago_last_error(void) { return lago::g_err; }
long long lago_reversed_launches(void) { return lago::g_reversed_launches.load(); }
long long lago_path_launches(int path) {
    return path >= 0 && path < lago::LP_COUNT ? lago::g_path_launches[path].load() : -1;
}
}
