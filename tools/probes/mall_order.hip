// Probe: does the 256 MB Infinity Cache reward a consumer that walks its input in the REVERSE of the order the producer
// wrote it?  Chain of out-of-place streaming kernels y = 2 x over `MB`-sized buffers (ping-pong), every kernel forward,
// against alternating direction (kernel i+1 starts where kernel i ended).  XCD-contiguous block order as in the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k(float4 *__restrict__ y, const float4 *__restrict__ x, unsigned total, int rev, int per) {
    unsigned b = blockIdx.x, q = total >> 3;
    unsigned L = b >= (q << 3) ? b : (b & 7u) * q + (b >> 3);
    if (rev) L = total - 1 - L;
    const size_t base = (size_t)L * per * 256;
    for (int i = 0; i < per; ++i) {
        const size_t j = base + (size_t)(rev ? per - 1 - i : i) * 256 + threadIdx.x;
        float4 v = x[j];
        v.x *= 2.f; v.y *= 2.f; v.z *= 2.f; v.w *= 2.f;
        y[j] = v;
    }
}

int main(int argc, char **argv) {
    for (int mb : {64, 128, 256, 400, 805, 1600}) {
        const size_t n4 = (size_t)mb * 1000000 / 16;
        const int per = 8;
        const unsigned total = (unsigned)(n4 / (256 * per));
        float4 *a, *b;
        CK(hipMalloc(&a, (size_t)total * per * 256 * 16)); CK(hipMalloc(&b, (size_t)total * per * 256 * 16));
        CK(hipMemset(a, 0, (size_t)total * per * 256 * 16));
        for (int alt = 0; alt < 2; ++alt) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            const int iters = 40;
            for (int w = 0; w < 2; ++w) {
                if (w) CK(hipEventRecord(e0));
                for (int i = 0; i < iters; ++i)
                    hipLaunchKernelGGL(k, dim3(total), dim3(256), 0, 0, (i & 1) ? a : b, (i & 1) ? b : a, total, alt ? (i & 1) : 0, per);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double bytes = 2.0 * total * per * 256 * 16;
            printf("%5d MB buffers, %s: %7.1f us per kernel  %6.2f TB/s (read + write)\n", mb, alt ? "alternating direction" : "always forward      ",
                   ms * 1e3 / iters, bytes / (ms * 1e-3 / iters) / 1e12);
        }
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
