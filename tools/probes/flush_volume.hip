// Probe (round 5): how much would a splat gain from FLUSHING FEWER CELLS?  The sheared-window splat flushes
// (TX + 1)(TY + 1) / (TX TY) window cells per source voxel (1.41 at its 5 x 6 x nz tiles, 1.56 at the 4 x 4 tiles of the
// three-channel form) with global float atomics, which run at 1.33 TB/s chip-wide.  A workgroup that MARCHES along x over
// a strip of TY rows -- a ring of window planes in LDS, a plane flushed when the march has passed it -- would flush
// (TY + 1) / TY x (XS + 1) / XS cells per voxel (XS = planes per workgroup): 1.2 at TY = 8, XS = 16.  This probe has the
// memory traffic of both schemes and nothing else (no positions, no LDS adds): per workgroup, XS times {stream one plane
// of TY x S voxels: NIN input planes, 3 output planes; flush TY + 1 rows of S cells into each of NC channel planes},
// plus one more flush at the end.  MARCH = 0: the one-shot tile (stream everything, then flush everything).
//   hipcc --offload-arch=gfx950 -O3 flush_volume.hip -o flush_volume
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int NT = 1024;
struct P { int S, B, XS, TY, NC, NIN, march, ntx, nty; };

__global__ __launch_bounds__(NT) void k(float* dI, float* out, const float* in, P p) {
    const int S = p.S;
    const size_t nv = (size_t)S * S * S;
    int b = blockIdx.x;
    const int n = b / (p.ntx * p.nty), r = b % (p.ntx * p.nty), bx = r / p.nty, by = r % p.nty;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float keep = 0.f;
    const int x0 = bx * p.XS, y0 = by * p.TY;
    auto stream_planes = [&](int xa, int xb) {
        const int per = p.TY * S;
        for (int t = threadIdx.x; t < (xb - xa) * per; t += NT) {
            const int a = xa + t / per, rr = t % per, gy = y0 + rr / S, kz = rr % S;
            const int gx = x0 + a;
            if (gx >= S || gy >= S) continue;
            const size_t sv = ((size_t)gx * S + gy) * S + kz;
            const float* q = in + (size_t)n * p.NIN * nv + sv;
            float acc = 0.f;
            for (int c = 0; c < p.NIN; ++c) acc += q[(size_t)c * nv];
            float* o = out + (size_t)n * 3 * nv + sv;
            o[0] = acc; o[nv] = acc * 2.f; o[2 * nv] = acc * 3.f;
            keep += acc;
        }
    };
    auto flush_planes = [&](int xa, int xb) {   // window planes xa .. xb-1 of this strip: TY + 1 rows each, NC channels
        const int rows = (xb - xa) * (p.TY + 1);
        for (int c = 0; c < p.NC; ++c)
            for (int row = wave; row < rows; row += NT / 64) {
                const int gx = min(x0 + xa + row / (p.TY + 1), S - 1), gy = min(y0 + row % (p.TY + 1), S - 1);
                float* grow = dI + ((size_t)n * p.NC + c) * nv + ((size_t)gx * S + gy) * S;
                for (int z = lane; z < S; z += 64) unsafeAtomicAdd(grow + z, 1.f + keep * 1e-30f);
            }
    };
    if (p.march) {
        for (int a = 0; a < p.XS; ++a) {
            stream_planes(a, a + 1);
            __syncthreads();
            flush_planes(a, a + 1);
        }
        flush_planes(p.XS, p.XS + 1);
    } else {
        stream_planes(0, p.XS);
        __syncthreads();
        flush_planes(0, p.XS + 1);
    }
    if (keep == 1.2345e30f) out[0] = 1.f;
}
int main(int argc, char** argv) {
    P p;
    p.S = argc > 1 ? atoi(argv[1]) : 128; p.B = 8;
    const size_t nv = (size_t)p.S * p.S * p.S;
    float *dI, *out, *in;
    (void)hipMalloc((void**)&dI, p.B * 3 * nv * 4); (void)hipMalloc((void**)&out, p.B * 3 * nv * 4); (void)hipMalloc((void**)&in, p.B * 9 * nv * 4);
    (void)hipMemset(dI, 0, p.B * 3 * nv * 4); (void)hipMemset(in, 0, p.B * 9 * nv * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int cfgs[][4] = {{5, 6, 1, 0}, {8, 8, 1, 0}, {16, 8, 1, 1}, {32, 8, 1, 1}, {16, 16, 1, 1}, {8, 8, 1, 1},
                           {4, 4, 3, 0}, {4, 6, 3, 0}, {16, 8, 3, 1}, {32, 8, 3, 1}, {16, 16, 3, 1}, {16, 4, 3, 1}};
    printf("S = %d, batch 8.  C = 1: 5 planes in, 3 out (32 B/voxel); C = 3: 9 in, 3 out (48 B/voxel)\n", p.S);
    for (auto& c : cfgs) {
        p.XS = c[0]; p.TY = c[1]; p.NC = c[2]; p.march = c[3]; p.NIN = p.NC == 1 ? 5 : 9;
        p.ntx = (p.S + p.XS - 1) / p.XS; p.nty = (p.S + p.TY - 1) / p.TY;
        const int blocks = p.B * p.ntx * p.nty;
        for (int i = 0; i < 3; ++i) k<<<blocks, NT>>>(dI, out, in, p);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        for (int i = 0; i < 10; ++i) k<<<blocks, NT>>>(dI, out, in, p);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        const double cells = (double)(p.XS + 1) * (p.TY + 1) / (p.XS * p.TY);
        printf("C = %d  %s %2d x %2d x %d  %5d workgroups  %.2f cells/voxel (%4.0f MB of atomics)  %7.1f us\n", p.NC,
               p.march ? "march" : "tile ", p.XS, p.TY, p.S, blocks, cells, cells * p.B * nv * 4 * p.NC / 1e6, ms / 10 * 1e3);
    }
    return 0;
}
