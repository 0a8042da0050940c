// Probe: the floor of the "tile -> float64 LDS window -> atomic flush" scheme with everything else stripped:
// coalesced loads of u (3 planes) and g, eight ds_add_f64 per voxel at the identity footprint, barrier, flush of the
// touched rows with global float atomics.  No position math, no weights, no gathers.  8 x 128^3 voxels.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int S = 128, B = 8, TX = 4, TY = 8, WX = 7, WY = 11, WZ = 128;
template <int MODE, int NT>  // bit0: LDS adds, bit1: flush atomics, bit2: loads
__global__ __launch_bounds__(NT) void k(float* dI, const float* u, const float* g) {
    extern __shared__ double win[];
    const int tiles_x = S / TX, tiles_y = S / TY;
    const int b = blockIdx.x;
    const int n = b / (tiles_x * tiles_y), r = b % (tiles_x * tiles_y), bx = r / tiles_y, by = r % tiles_y;
    const size_t nv = (size_t)S * S * S;
    for (int f = threadIdx.x; f < WX * WY * WZ; f += NT) win[f] = 0.0;
    __syncthreads();
    for (int t = threadIdx.x; t < TX * TY * S; t += NT) {
        const int a = t / (TY * S), rr = t % (TY * S), c = rr / S, kz = rr % S;
        const size_t sv = ((size_t)(bx * TX + a) * S + (by * TY + c)) * S + kz;
        float w = 1.f;
        if (MODE & 4) w = u[n * 3 * nv + sv] + u[n * 3 * nv + nv + sv] + u[n * 3 * nv + 2 * nv + sv] + g[n * nv + sv];
        if (MODE & 1) {
            const int z1 = kz + 1 < S ? kz + 1 : kz;
            double* w0 = win + ((a + 1) * WY + (c + 1)) * WZ;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                double* row = w0 + (q >> 1) * WY * WZ + (q & 1) * WZ;
                __hip_atomic_fetch_add(row + kz, (double)w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(row + z1, (double)(w * 0.5f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else if (w == 1.2345e30f) win[0] = 1.0;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = wave; row < WX * WY; row += NT / 64) {
        const int lx = row / WY, ly = row % WY;
        const int gx = bx * TX + lx - 1, gy = by * TY + ly - 1;
        if (gx < 0 || gy < 0 || gx >= S || gy >= S) continue;
        float* grow = dI + n * nv + ((size_t)gx * S + gy) * S;
        for (int z = lane; z < WZ; z += 64) {
            const double acc = win[row * WZ + z];
            if (acc != 0.0 && (MODE & 2)) unsafeAtomicAdd(grow + z, (float)acc);
            else if (acc == 1.2345e300) grow[z] = 1.f;
        }
    }
}
template <int MODE, int NT> float run(float* dI, const float* u, const float* g) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int blocks = B * (S / TX) * (S / TY);
    const size_t smem = WX * WY * WZ * 8;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    for (int i = 0; i < 3; ++i) k<MODE, NT><<<blocks, NT, smem>>>(dI, u, g);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < 10; ++i) k<MODE, NT><<<blocks, NT, smem>>>(dI, u, g);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 10;
}
int main() {
    const size_t nv = (size_t)S * S * S;
    float *dI, *u, *g;
    (void)hipMalloc(&dI, B * nv * 4); (void)hipMalloc(&u, 3 * B * nv * 4); (void)hipMalloc(&g, B * nv * 4);
    (void)hipMemset(dI, 0, B * nv * 4); (void)hipMemset(u, 0, 3 * B * nv * 4); (void)hipMemset(g, 0, B * nv * 4);
    printf("tile 4x8x128, window 7x11x128 f64 (77 KB), 8 x 128^3 voxels; us per launch\n");
    printf("NT=512  loads+adds+flush %.1f | adds+flush %.1f | loads+adds %.1f | loads+flush(no adds) %.1f | loads only %.1f | adds only %.1f | nothing %.1f\n",
           1e3 * run<7, 512>(dI, u, g), 1e3 * run<3, 512>(dI, u, g), 1e3 * run<5, 512>(dI, u, g), 1e3 * run<6, 512>(dI, u, g),
           1e3 * run<4, 512>(dI, u, g), 1e3 * run<1, 512>(dI, u, g), 1e3 * run<0, 512>(dI, u, g));
    printf("NT=1024 loads+adds+flush %.1f | adds+flush %.1f | loads+adds %.1f | loads only %.1f | adds only %.1f | nothing %.1f\n",
           1e3 * run<7, 1024>(dI, u, g), 1e3 * run<3, 1024>(dI, u, g), 1e3 * run<5, 1024>(dI, u, g), 1e3 * run<4, 1024>(dI, u, g),
           1e3 * run<1, 1024>(dI, u, g), 1e3 * run<0, 1024>(dI, u, g));
    printf("NT=256  loads+adds+flush %.1f | adds+flush %.1f | loads+adds %.1f | loads only %.1f | adds only %.1f | nothing %.1f\n",
           1e3 * run<7, 256>(dI, u, g), 1e3 * run<3, 256>(dI, u, g), 1e3 * run<5, 256>(dI, u, g), 1e3 * run<4, 256>(dI, u, g),
           1e3 * run<1, 256>(dI, u, g), 1e3 * run<0, 256>(dI, u, g));
    return 0;
}
