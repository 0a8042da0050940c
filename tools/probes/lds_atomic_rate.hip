// Probe: throughput of LDS float atomics (ds_add_f32) vs integer atomics and plain LDS RMW on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    unsigned long long* l64 = reinterpret_cast<unsigned long long*>(lds);
    double* d64 = reinterpret_cast<double*>(lds);
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 0;
    __syncthreads();
    if (MODE == 7 || MODE == 8)  // MODE register, FP_DENORM single-precision field (bits 4:5) := 0 (flush in and out)
        __builtin_amdgcn_s_setreg(1 | (4 << 6) | (1 << 11), 0);
    int base = threadIdx.x;  // conflict-free: consecutive lanes, consecutive dwords
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            int idx = (base + q * 256 + it * 64) & 8191;
            if (MODE == 0 || MODE == 7) __hip_atomic_fetch_add(&lds[idx], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 1) __hip_atomic_fetch_add((int*)&lds[idx], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 2) lds[idx] += 1.0f;  // non-atomic RMW (racy across waves; rate only)
            if (MODE == 4) __hip_atomic_fetch_add(&l64[idx & 4095], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 5) __hip_atomic_fetch_add(&d64[idx & 4095], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 6) { float old = __hip_atomic_fetch_add(&lds[idx], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); if (old == 12345.f) out[1] = old; }
            if (MODE == 8) __hip_atomic_fetch_add(&lds[idx], 1.0e-3f * (float)(q + it), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 3) { int i4 = (base * 4 + q * 1024 + it * 256) & 8191;  // stride-4 lanes (VPL=4 pattern)
                             __hip_atomic_fetch_add(&lds[i4], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[5];
}
template <int MODE> float run(float* o, int blocks, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(o, iters); hipDeviceSynchronize();
    hipEventRecord(a); k<MODE><<<blocks, 256>>>(o, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    float* o; hipMalloc(&o, 1 << 20);
    const int blocks = 256 * 8, iters = 512;
    const double waveops = (double)blocks * 4 * iters * 8;  // wave-level atomic instructions
    const char* names[] = {"ds_add_f32 (conflict-free)", "ds_add_u32 (conflict-free)", "plain RMW", "ds_add_f32 stride-4 lanes", "ds_add_u64", "ds_add_f64", "ds_add_rtn_f32", "ds_add_f32, f32 denormals flushed (MODE)", "same, varying addends"};
    float ms[9] = {run<0>(o, blocks, iters), run<1>(o, blocks, iters), run<2>(o, blocks, iters), run<3>(o, blocks, iters), run<4>(o, blocks, iters), run<5>(o, blocks, iters), run<6>(o, blocks, iters), run<7>(o, blocks, iters), run<8>(o, blocks, iters)};
    for (int m = 0; m < 9; ++m)
        printf("%-28s %8.3f ms  -> %6.1f cycles per wave-instruction per CU (2.4 GHz, 256 CUs)\n", names[m], ms[m],
               ms[m] * 1e-3 * 2.4e9 / (waveops / 256));
    return 0;
}
