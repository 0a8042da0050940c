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
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) return fail_hip(e, what);
    }
    return LAGO_OK;
}

}  // namespace lago

extern "C" {
void lago_set_debug(int on) { lago::g_debug = on ? 1 : 0; }
int lago_get_debug(void) { return lago::g_debug; }
int lago_abi_version(void) { return LAGO_ABI_VERSION; }
const char *lago_version(void) { return "lagomorph_hip 0.1 (gfx950, HIP)"; }
const char *lago_last_error(void) { return lago::g_err; }
void lago_set_splat_mode(int mode) { lago::g_splat_mode = mode; }
int lago_get_splat_mode(void) { return lago::g_splat_mode; }
void lago_set_vector_kernels(int on) { lago::g_interp_vec = on ? 1 : 0; }
long long lago_path_launches(int path) {
    return path >= 0 && path < lago::LP_COUNT ? lago::g_path_launches[path].load() : -1;
}
void lago_set_launch_order(int alternate) { lago::g_launch_alt = alternate ? 1 : 0; }
}
