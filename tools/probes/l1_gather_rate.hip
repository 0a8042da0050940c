// Probe: L1 (TCP) throughput of the access shapes a trilinear gather can use, on an L1-resident buffer.
// Each wave issues `iters` loads of one shape; lanes are z-consecutive (lane stride 4 bytes) plus a
// per-iteration row offset, as in the interp kernels.  Reports cycles per wave-level load per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const float *buf, float *out, int iters, int rowstride, int misalign) {
    // wave-uniform descriptor, per-lane byte offset
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(buf), 0, 1u << 20, 0x00020000);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned off = (unsigned)(lane * 4 + misalign * 4 + wave * 1024);
    float acc = 0.f;
    __shared__ float lbuf[4 * 4 * 64 * 4];   // LDS-direct shapes: [wave][load in flight][lane][up to 4 dwords]
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {   // four independent loads in flight per wave
        const unsigned o = off + (unsigned)((((i & 1) * 4 + j) & 7) * rowstride);
        if (SHAPE == 0) {  // one dword per lane, consecutive
            acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, o, 0, 0));
        } else if (SHAPE == 1) {  // overlapping pairs: 8 bytes per lane at a 4-byte lane stride
            unsigned long long v = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(r, o, 0, 0));
            acc += __builtin_bit_cast(float, (unsigned)v) + __builtin_bit_cast(float, (unsigned)(v >> 32));
        } else if (SHAPE == 2) {  // two dword loads (lo, hi)
            acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, o, 0, 0));
            acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, o + 4, 0, 0));
        } else if (SHAPE == 3) {  // aligned pairs: 8 bytes per lane at an 8-byte lane stride
            unsigned long long v = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(r, o + lane * 4, 0, 0));
            acc += __builtin_bit_cast(float, (unsigned)v) + __builtin_bit_cast(float, (unsigned)(v >> 32));
        } else if (SHAPE == 4) {  // 16 bytes per lane at a 16-byte stride (streaming shape)
            u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (o & ~15u) + lane * 12, 0, 0);
            acc += __builtin_bit_cast(float, v.x) + __builtin_bit_cast(float, v.w);
        } else if (SHAPE == 5) {  // dword per lane, every other lane active (fallback-style masked load)
            if (lane & 1) acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, o, 0, 0));
        } else if (SHAPE == 7) {  // global_load_dwordx2 (64-bit address per lane) at a 4-byte lane stride
            typedef unsigned long long ull4 __attribute__((aligned(4)));
            const ull4 v = *reinterpret_cast<const ull4 *>(reinterpret_cast<const char *>(buf) + o);
            acc += __builtin_bit_cast(float, (unsigned)v) + __builtin_bit_cast(float, (unsigned)(v >> 32));
        } else if (SHAPE == 8) {  // global_load_dword, consecutive lanes
            acc += *reinterpret_cast<const float *>(reinterpret_cast<const char *>(buf) + o);
        } else if (SHAPE == 9) {  // dwordx2 buffer load, every lane the SAME pair as its neighbour pair-wise (lanes 2k, 2k+1 share)
            unsigned long long v = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(r, (o & ~7u) - (lane & 1) * 4 + 0, 0, 0));
            acc += __builtin_bit_cast(float, (unsigned)v) + __builtin_bit_cast(float, (unsigned)(v >> 32));
        } else if (SHAPE == 10) {  // LDS-direct: global_load_lds_dword, consecutive lanes (lands at M0 + 4 lane)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(buf) + o),
                                             (__attribute__((address_space(3))) void *)(lbuf + (wave * 4 + j) * 256), 4, 0, 0);
        } else if (SHAPE == 11) {  // LDS-direct: global_load_lds_dwordx4 (gfx950), 16 B per lane at a 16-byte stride
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(buf) + (o & ~15u) + lane * 12),
                                             (__attribute__((address_space(3))) void *)(lbuf + (wave * 4 + j) * 256), 16, 0, 0);
        } else if (SHAPE == 12) {  // LDS-direct dword gather + reading the value back from LDS (the full round trip)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(buf) + o),
                                             (__attribute__((address_space(3))) void *)(lbuf + (wave * 4 + j) * 256), 4, 0, 0);
            if (j == 3) {
                __builtin_amdgcn_s_waitcnt(0);
                acc += lbuf[(wave * 4 + 0) * 256 + lane] + lbuf[(wave * 4 + 1) * 256 + lane] + lbuf[(wave * 4 + 2) * 256 + lane] +
                       lbuf[(wave * 4 + 3) * 256 + lane];
            }
        } else if (SHAPE == 13) {  // two GENUINE dword loads lo / hi: the second offset is opaque, so the compiler cannot
                                   // merge the pair into one dwordx2 (which is what happens to SHAPE 2)
            unsigned o2 = o + 4u;
            asm volatile("" : "+v"(o2));
            acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, o, 0, 0));
            acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, o2, 0, 0));
        } else if (SHAPE == 14) {  // genuine dword loads with a drifting lane stride (a smooth displacement: the sample
                                   // index advances by 1.1 cells per lane) -- the gather's real address pattern
            const unsigned og = o + (unsigned)((lane / 10) * 4);
            acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, og, 0, 0));
        } else if (SHAPE == 15) {  // dwordx2 with the same drifting pattern
            const unsigned og = o + (unsigned)((lane / 10) * 4);
            unsigned long long v = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(r, og, 0, 0));
            acc += __builtin_bit_cast(float, (unsigned)v) + __builtin_bit_cast(float, (unsigned)(v >> 32));
        } else if (SHAPE == 6) {  // dword per lane, one lane in four active
            if ((lane & 3) == 0) acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, o, 0, 0));
        }
      }
    }
    if (SHAPE >= 10 && SHAPE <= 11) {
        __builtin_amdgcn_s_waitcnt(0);
        acc += lbuf[threadIdx.x];
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int SHAPE>
static void run(const char *name, const float *buf, float *out, int rowstride, int misalign) {
    const int iters = 1024, blocks = 256 * 8;  // 8 workgroups of 4 waves per CU
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, buf, out, iters, rowstride, misalign);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, buf, out, iters, rowstride, misalign);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double loads_per_cu = (double)blocks / 256 * 4 * iters * 4 * (SHAPE == 2 || SHAPE == 13 ? 2 : 1);
    printf("%-58s rowstride %4d misalign %d: %7.3f ms  %6.2f ns per wave-load per CU (%5.1f clk @2.4GHz)\n", name, rowstride,
           misalign, ms, ms * 1e6 / loads_per_cu, ms * 1e6 / loads_per_cu * 2.4);
}

int main() {
    float *buf, *out;
    CK(hipMalloc(&buf, 1 << 20)); CK(hipMalloc(&out, 64)); CK(hipMemset(buf, 0, 1 << 20));
    for (int mis = 0; mis < 2; ++mis) {
        run<0>("dword, consecutive lanes (256 B per wave)", buf, out, 512, mis);
        run<1>("dwordx2 at 4 B lane stride (overlapping pairs, 260 B span)", buf, out, 512, mis);
        run<2>("two dwords lo/hi (each counted)", buf, out, 512, mis);
        run<3>("dwordx2 at 8 B lane stride (512 B per wave)", buf, out, 512, mis);
        run<4>("dwordx4 at 12 B lane stride", buf, out, 512, mis);
        run<5>("dword, every other lane", buf, out, 512, mis);
        run<6>("dword, one lane in four", buf, out, 512, mis);
        run<7>("global_load_dwordx2 at 4 B lane stride", buf, out, 512, mis);
        run<8>("global_load_dword, consecutive lanes", buf, out, 512, mis);
        run<9>("dwordx2, 8-byte aligned, lanes pairwise overlapping", buf, out, 512, mis);
        run<13>("two genuine dword loads lo/hi (each counted)", buf, out, 512, mis);
        run<14>("dword, drifting lane stride (1.1 cells per lane)", buf, out, 512, mis);
        run<15>("dwordx2, drifting lane stride", buf, out, 512, mis);
        run<10>("LDS-direct global_load_lds_dword, consecutive lanes", buf, out, 512, mis);
        run<11>("LDS-direct global_load_lds_dwordx4, 16 B stride", buf, out, 512, mis);
        run<12>("LDS-direct dword + ds_read back", buf, out, 512, mis);
    }
    return 0;
}
