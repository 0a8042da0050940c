// Probe: HBM throughput of segment-strided access patterns (layout study for the fluid-metric FFT passes).
// Every workgroup moves `nseg` segments of `lps` float2 each through LDS:
//   address(wg, seg) = (wg / wgi) * wgo + (wg % wgi) * wgs + (seg / si) * so + (seg % si) * ss   (in float2 units)
// mode 0: read then write in place; 1: read only; 2: write only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Pat { long long wgi, wgo, wgs, si, so, ss; int nseg, lps, mode; };

template <int K>
__global__ __launch_bounds__(256) void seg_kernel(float2 *buf, Pat p, float *sink) {
    extern __shared__ float2 lds[];
    const long long wg = blockIdx.x;
    float2 *base = buf + (wg / p.wgi) * p.wgo + (wg % p.wgi) * p.wgs;
    const int n = p.nseg * p.lps;
    if (p.mode != 2) {
        // all K loads of a thread are issued before the first LDS store
        float2 v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int i = threadIdx.x + k * 256;
            const int ii = i < n ? i : 0;
            const int seg = ii / p.lps, lane = ii - seg * p.lps;
            v[k] = base[(seg / p.si) * p.so + (seg % p.si) * p.ss + lane];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int i = threadIdx.x + k * 256;
            if (i < n) lds[i] = v[k];
        }
    } else {
        for (int i = threadIdx.x; i < n; i += 256) lds[i] = make_float2((float)i, 1.f);
    }
    __syncthreads();
    if (p.mode != 1) {
        for (int i = threadIdx.x; i < n; i += 256) {
            const int seg = i / p.lps, lane = i - seg * p.lps;
            float2 v = lds[(i + 64) % n];
            base[(seg / p.si) * p.so + (seg % p.si) * p.ss + lane] = v;
        }
    } else {
        float acc = 0;
        for (int i = threadIdx.x; i < n; i += 256) acc += lds[(i + 64) % n].x;
        if (acc == 12345.678f) sink[0] = acc;
    }
}

static void run(const char *name, float2 *buf, float *sink, Pat p, long long nwg, double bytes_per_wg) {
    size_t smem = (size_t)p.nseg * p.lps * sizeof(float2);
    const int K = (p.nseg * p.lps + 255) / 256;
    auto kern = K <= 20 ? seg_kernel<20> : K <= 24 ? seg_kernel<24> : K <= 34 ? seg_kernel<34> : seg_kernel<48>;
    if (K > 48) { printf("K too large\n"); return; }
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int mode = 0; mode < 3; ++mode) {
        p.mode = mode;
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), smem, 0, buf, p, sink);
        CK(hipEventRecord(a));
        const int reps = 5;
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), smem, 0, buf, p, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        ms /= reps;
        const double bytes = bytes_per_wg * nwg * (mode == 0 ? 2 : 1);
        printf("%-44s mode %d (%s): %7.3f ms  %7.1f GB/s\n", name, mode, mode == 0 ? "r+w" : mode == 1 ? "read" : "write", ms, bytes / ms / 1e6);
    }
}

int main() {
    // spectrum of 32 x 3 x 128 x 128 x 65 complex
    const long long NN = 32, C = 3, NX = 128, NY = 128, NZC = 65;
    const long long total = NN * C * NX * NY * NZC;
    float2 *buf;
    float *sink;
    CK(hipMalloc(&buf, (total + 4096) * sizeof(float2)));
    CK(hipMalloc(&sink, 16));
    CK(hipMemset(buf, 0, (total + 4096) * sizeof(float2)));
    Pat p;
    // A: current x-pass.  wg = (n, y, chunk[5 of 13]); seg = (c, x): stride NY*NZC; 13 lanes
    p = {NY * 5, C * NX * NY * NZC, 13, 1, NY * NZC, 0, (int)(C * NX), 13, 0};
    // wg % (NY*5) -> y*5+chunk: address y*NZC + chunk*13 = (y*5+chunk)*13 since NZC = 65 = 5*13
    run("A x-pass now: 104B seg @ 66560B stride", buf, sink, p, NN * NY * 5, C * NX * 13 * 8.0);
    // B: layout [n][ky][c][x][kz]: wg = (n, ky, chunk): base (n*NY+ky)*C*NX*NZC + chunk*13; seg = (c,x): stride NZC
    p = {5, C * NX * NZC, 13, 1, NZC, 0, (int)(C * NX), 13, 0};
    run("B [n][ky][c][x][kz]: 104B seg @ 520B stride", buf, sink, p, NN * NY * 5, C * NX * 13 * 8.0);
    // C: fully contiguous 39 KB per wg
    p = {1, C * NX * 13, 0, 1, 13, 0, (int)(C * NX), 13, 0};
    run("C contiguous 39KB per wg", buf, sink, p, NN * NY * 5, C * NX * 13 * 8.0);
    // C16: contiguous 48 KB per wg with 16 lanes (128 B segments)
    p = {1, C * NX * 16, 0, 1, 16, 0, (int)(C * NX), 16, 0};
    run("C16 contiguous 48KB per wg", buf, sink, p, NN * NY * 4, C * NX * 16 * 8.0);
    // D: zy pass, natural layout: wg = (n,c,x): contiguous NY*NZC plane
    p = {1, NY * NZC, 0, 1, NZC, 0, (int)NY, (int)NZC, 0};
    run("D zy plane contiguous 66.5KB", buf, sink, p, NN * C * NX, NY * NZC * 8.0);
    // E: zy pass writing layout B: wg = (n, c, x) -> base n*(NY*C*NX*NZC) + c*NX*NZC + x*NZC; seg = ky: stride C*NX*NZC
    p = {C * NX, NY * C * NX * NZC, NZC, 1, C * NX * NZC, 0, (int)NY, (int)NZC, 0};
    run("E zy rows 520B @ 199680B stride", buf, sink, p, NN * C * NX, NY * NZC * 8.0);
    // F: zy pass writing chunked layout [n][ky][chunk(4)][c][x][16] (main 64 columns only):
    //    wg = (n,c,x): base n*(NY*4*C*NX*16) + c*NX*16 + x*16 ; seg = (ky, chunk): stride C*NX*16
    p = {C * NX, NY * 4 * C * NX * 16, 16, 1, C * NX * 16, 0, (int)(NY * 4), 16, 0};
    run("F zy 128B seg @ 49152B stride (chunked)", buf, sink, p, NN * C * NX, NY * 64 * 8.0);
    // G: like A but 16 lanes/128 B aligned segments on a 64-column spectrum (pitch 64)
    p = {NY * 4, C * NX * NY * 64, 16, 1, NY * 64, 0, (int)(C * NX), 16, 0};
    run("G x-pass 128B aligned seg @ 65536B stride", buf, sink, p, NN * NY * 4, C * NX * 16 * 8.0);
    // H: x-pass with 256 B segments (32 lanes) at 66560 stride
    p = {NY * 2, C * NX * NY * NZC, 32, 1, NY * NZC, 0, (int)(C * NX), 32, 0};
    // wg%(NY*2) -> y*2+half: want y*NZC + half*32; approximate with 32.5 -> use wgs=32 (overlaps a little, fine for a probe)
    run("H x-pass 256B seg @ 66560B stride (approx)", buf, sink, p, NN * NY * 2, C * NX * 32 * 8.0);
    return 0;
}
