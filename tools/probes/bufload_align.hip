// Probe: do buffer_load_dwordx2 / global_load_dwordx2 honour 4-byte (not 8-byte) aligned addresses on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) F2 { float a, b; };
__global__ void k(float* out, const float* img, const int* flags, int n) {
    int i = threadIdx.x;
    if (i >= n - 1) return;
    auto r = __builtin_amdgcn_make_buffer_rsrc((void*)img, 0, (unsigned)(n * 4), 0x00020000);
    u32x2 p = __builtin_amdgcn_raw_buffer_load_b64(r, (unsigned)(i * 4), 0, 0);
    unsigned long long q = __builtin_bit_cast(unsigned long long, p);
    float lo = __builtin_bit_cast(float, (unsigned)q), hi = __builtin_bit_cast(float, (unsigned)(q >> 32));
    bool f_hi = flags[i] & 1, c_lo = flags[i] & 2;   // both false on the host side: expect (lo, hi)
    out[4 * i + 0] = f_hi ? hi : lo;
    out[4 * i + 1] = c_lo ? lo : hi;
    F2 g = *reinterpret_cast<const F2*>(reinterpret_cast<const char*>(img) + i * 4);
    out[4 * i + 2] = g.a;
    out[4 * i + 3] = g.b;
}
int main() {
    const int n = 64;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, n * 16);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(o, 0, n * 16);
    int* fl; hipMalloc(&fl, n*4); hipMemset(fl, 0, n*4);
    k<<<1, 64>>>(o, d, fl, n);
    std::vector<float> r(n * 4);
    hipMemcpy(r.data(), o, n * 16, hipMemcpyDeviceToHost);
    int badb = 0, badg = 0;
    for (int i = 0; i < n - 1; ++i) {
        if (r[4*i] != i || r[4*i+1] != i + 1) { if (badb < 4) printf("buffer i=%d got (%g,%g)\n", i, r[4*i], r[4*i+1]); badb++; }
        if (r[4*i+2] != i || r[4*i+3] != i + 1) { if (badg < 4) printf("global i=%d got (%g,%g)\n", i, r[4*i+2], r[4*i+3]); badg++; }
    }
    printf("buffer_load_dwordx2 unaligned: %d bad; global_load_dwordx2 unaligned: %d bad\n", badb, badg);
    return 0;
}
