// Probe: does a trilinear gather through an LDS window beat the pair-gather form?
//
// compose-like op on 32 x 3 x 128^3 float: out_c = ds*u_c + dt * v_c(x + u(x)), c = 0..2.
//   K0  the shipped shape: 256 threads, 2 voxels per lane, four dwordx2 pair gathers per channel (TCP)
//   K1  window form: 1024 threads own an 8 x 16 x 32 tile; per channel the tile's source window
//       (tile + halo) goes global -> LDS with global_load_lds_dwordx4, the eight corners come from LDS;
//       lanes whose corners leave the window fall back to global loads.
// Both compute the same expression in the same order; the outputs are compared.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int NX = 128, NY = 128, NZ = 128;
constexpr size_t NV = (size_t)NX * NY * NZ;

struct Pt {
    int fx, cx, fy, cy, fz, cz;
    float tx, ty, tz;
};

__device__ __forceinline__ Pt setup(int i, int j, int k, float ux, float uy, float uz) {
    Pt p;
    const float x = (float)i + ux, y = (float)j + uy, z = (float)k + uz;
    const float flx = floorf(x), fly = floorf(y), flz = floorf(z);
    p.tx = x - flx; p.ty = y - fly; p.tz = z - flz;
    p.fx = (int)flx; p.fy = (int)fly; p.fz = (int)flz;
    p.cx = p.fx + 1; p.cy = p.fy + 1; p.cz = p.fz + 1;
    p.fx = min(max(p.fx, 0), NX - 1); p.cx = min(max(p.cx, 0), NX - 1);
    p.fy = min(max(p.fy, 0), NY - 1); p.cy = min(max(p.cy, 0), NY - 1);
    p.fz = min(max(p.fz, 0), NZ - 1); p.cz = min(max(p.cz, 0), NZ - 1);
    return p;
}

__device__ __forceinline__ float lerp8(const Pt &p, float c000, float c001, float c010, float c011, float c100, float c101,
                                       float c110, float c111) {
    const float a00 = c000 + p.tz * (c001 - c000), a01 = c010 + p.tz * (c011 - c010);
    const float a10 = c100 + p.tz * (c101 - c100), a11 = c110 + p.tz * (c111 - c110);
    const float b0 = a00 + p.ty * (a01 - a00), b1 = a10 + p.ty * (a11 - a10);
    return b0 + p.tx * (b1 - b0);
}

typedef unsigned long long ull4 __attribute__((aligned(4)));

// ---------------------------------------------------------------- K0: pair gathers through the TCP
template <int U>
__global__ __launch_bounds__(256) void k_pair(float *__restrict__ out, const float *__restrict__ u,
                                              const float *__restrict__ v, float ds, float dt) {
    const unsigned nbx = (unsigned)(NV / (256 * U));
    const unsigned n = blockIdx.x / nbx, bx = blockIdx.x - n * nbx;
    const float *un = u + (size_t)n * 3 * NV, *vn = v + (size_t)n * 3 * NV;
    float *on = out + (size_t)n * 3 * NV;
    unsigned s[U];
    float uu[3][U];
    Pt p[U];
    unsigned o00[U], o01[U], o10[U], o11[U];
    bool hi[U];
#pragma unroll
    for (int e = 0; e < U; ++e) {
        s[e] = (bx * U + e) * 256 + threadIdx.x;
#pragma unroll
        for (int d = 0; d < 3; ++d) uu[d][e] = un[(size_t)d * NV + s[e]];
    }
#pragma unroll
    for (int e = 0; e < U; ++e) {
        const int i = s[e] / (NY * NZ), r = s[e] - i * (NY * NZ), j = r / NZ, k = r - j * NZ;
        p[e] = setup(i, j, k, uu[0][e], uu[1][e], uu[2][e]);
        const int zb = min(p[e].fz, NZ - 2);  // pair base: (zb, zb + 1) always inside the row
        hi[e] = p[e].fz > zb;                 // fz == cz == NZ - 1: both corners are the pair's high word
        o00[e] = (p[e].fx * NY + p[e].fy) * NZ + zb;
        o01[e] = (p[e].fx * NY + p[e].cy) * NZ + zb;
        o10[e] = (p[e].cx * NY + p[e].fy) * NZ + zb;
        o11[e] = (p[e].cx * NY + p[e].cy) * NZ + zb;
        if (p[e].cz == p[e].fz && !hi[e]) p[e].tz = 0.f;  // clamped low end
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float *vc = vn + (size_t)c * NV;
        float o[U];
#pragma unroll
        for (int e = 0; e < U; ++e) {
            const ull4 q00 = *reinterpret_cast<const ull4 *>(vc + o00[e]);
            const ull4 q01 = *reinterpret_cast<const ull4 *>(vc + o01[e]);
            const ull4 q10 = *reinterpret_cast<const ull4 *>(vc + o10[e]);
            const ull4 q11 = *reinterpret_cast<const ull4 *>(vc + o11[e]);
            auto lo = [&](ull4 q) { return __builtin_bit_cast(float, (unsigned)(hi[e] ? (q >> 32) : q)); };
            auto hh = [&](ull4 q) { return __builtin_bit_cast(float, (unsigned)(q >> 32)); };
            const float val = lerp8(p[e], lo(q00), hh(q00), lo(q01), hh(q01), lo(q10), hh(q10), lo(q11), hh(q11));
            o[e] = ds * uu[c][e] + dt * val;
        }
#pragma unroll
        for (int e = 0; e < U; ++e) on[(size_t)c * NV + s[e]] = o[e];
    }
}

// ---------------------------------------------------------------- K0p: pair gathers + an L2-warming wave
// Hypothesis (profiles/r03_tcp_counters.md): the vector L1 returns in order, so gather hits queue behind the misses of
// the streamed operands.  A fifth wave of every workgroup touches, through the SCALAR cache (a different path), the
// lines a later workgroup of the same XCD will stream -- u of block b + 8 D, and the leading-edge slab of v -- so that
// those become L2 hits.  Blocks go round-robin over the 8 XCDs, so b + 8 D shares b's L2.
template <int U>
__global__ __launch_bounds__(320) void k_pair_pf(float *__restrict__ out, const float *__restrict__ u,
                                                 const float *__restrict__ v, float ds, float dt, int D, int vslab) {
    if (threadIdx.x >= 256) {
        if (D <= 0) return;
        const unsigned nbx = (unsigned)(NV / (256 * U));
        const unsigned tb = blockIdx.x + 8u * (unsigned)D;
        if (tb >= gridDim.x) return;
        const unsigned n = tb / nbx, bx = tb - n * nbx;
        // 512 voxels x 3 channels of u: 16 lines of 128 B per channel; v the same range `vslab` slabs ahead
        const char *ub = reinterpret_cast<const char *>(u + (size_t)n * 3 * NV + (size_t)bx * 256 * U);
        const char *vb = reinterpret_cast<const char *>(v + (size_t)n * 3 * NV + (size_t)bx * 256 * U + (size_t)vslab * NY * NZ);
        const bool vok = bx * 256 * U + (size_t)vslab * NY * NZ + 256 * U <= NV;
        int acc = 0;
#pragma unroll 1
        for (int c = 0; c < 3; ++c) {
            int t0[16], t1[16];
#pragma unroll
            for (int l = 0; l < 16; ++l) {  // 32 scalar loads in flight, then one wait
                const char *pu = ub + (size_t)c * NV * 4 + l * 128;
                asm volatile("s_load_dword %0, %1, 0x0" : "=s"(t0[l]) : "s"(pu) : "memory");
                const char *pv = (vok ? vb : ub) + (size_t)c * NV * 4 + l * 128;
                asm volatile("s_load_dword %0, %1, 0x0" : "=s"(t1[l]) : "s"(pv) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int l = 0; l < 16; ++l) acc += t0[l] + t1[l];
        }
        if (acc == 0x7fffffff) out[0] = 0.f;  // keep the loads
        return;
    }
    const unsigned nbx = (unsigned)(NV / (256 * U));
    const unsigned n = blockIdx.x / nbx, bx = blockIdx.x - n * nbx;
    const float *un = u + (size_t)n * 3 * NV, *vn = v + (size_t)n * 3 * NV;
    float *on = out + (size_t)n * 3 * NV;
    unsigned s[U];
    float uu[3][U];
    Pt p[U];
    unsigned o00[U], o01[U], o10[U], o11[U];
    bool hi[U];
#pragma unroll
    for (int e = 0; e < U; ++e) {
        s[e] = (bx * U + e) * 256 + threadIdx.x;
#pragma unroll
        for (int d = 0; d < 3; ++d) uu[d][e] = un[(size_t)d * NV + s[e]];
    }
#pragma unroll
    for (int e = 0; e < U; ++e) {
        const int i = s[e] / (NY * NZ), r = s[e] - i * (NY * NZ), j = r / NZ, k = r - j * NZ;
        p[e] = setup(i, j, k, uu[0][e], uu[1][e], uu[2][e]);
        const int zb = min(p[e].fz, NZ - 2);
        hi[e] = p[e].fz > zb;
        o00[e] = (p[e].fx * NY + p[e].fy) * NZ + zb;
        o01[e] = (p[e].fx * NY + p[e].cy) * NZ + zb;
        o10[e] = (p[e].cx * NY + p[e].fy) * NZ + zb;
        o11[e] = (p[e].cx * NY + p[e].cy) * NZ + zb;
        if (p[e].cz == p[e].fz && !hi[e]) p[e].tz = 0.f;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float *vc = vn + (size_t)c * NV;
        float o[U];
#pragma unroll
        for (int e = 0; e < U; ++e) {
            const ull4 q00 = *reinterpret_cast<const ull4 *>(vc + o00[e]);
            const ull4 q01 = *reinterpret_cast<const ull4 *>(vc + o01[e]);
            const ull4 q10 = *reinterpret_cast<const ull4 *>(vc + o10[e]);
            const ull4 q11 = *reinterpret_cast<const ull4 *>(vc + o11[e]);
            auto lo = [&](ull4 q) { return __builtin_bit_cast(float, (unsigned)(hi[e] ? (q >> 32) : q)); };
            auto hh = [&](ull4 q) { return __builtin_bit_cast(float, (unsigned)(q >> 32)); };
            const float val = lerp8(p[e], lo(q00), hh(q00), lo(q01), hh(q01), lo(q10), hh(q10), lo(q11), hh(q11));
            o[e] = ds * uu[c][e] + dt * val;
        }
#pragma unroll
        for (int e = 0; e < U; ++e) on[(size_t)c * NV + s[e]] = o[e];
    }
}

// ---------------------------------------------------------------- K1: LDS window
constexpr int TX = 8, TY = 16, TZ = 32, H = 2;
constexpr int WX = TX + 1 + 2 * H, WY = TY + 1 + 2 * H;  // 13 x 21 rows
constexpr int ZLO = 4;                                   // z halo below (keeps 16-byte chunks aligned)
constexpr int WZC = 11, WZ = WZC * 4;                    // 44 floats per row: z0 - 4 .. z0 + 39
constexpr int NCHUNK = WX * WY * WZC;                    // 3003 16-byte chunks per channel
template <int NT1, int U1, int WPE, bool DB, bool FB>
__global__ __launch_bounds__(NT1) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_window(float *__restrict__ out, const float *__restrict__ u,
                                                const float *__restrict__ v, float ds, float dt,
                                                unsigned *__restrict__ nfallback) {
    extern __shared__ float win[];  // [DB ? 2 : 1][NCHUNK * 4 (+ slack to whole rounds)]
    constexpr int ROUNDS = (NCHUNK + NT1 - 1) / NT1;
    constexpr int WBUF = ROUNDS * NT1 * 4;
    constexpr int XS = NT1 / 512;  // x slices covered by one pass of the workgroup
    static_assert(XS * U1 == TX, "tile");
    constexpr unsigned tiles_z = NZ / TZ, tiles_y = NY / TY, tiles_x = NX / TX, tiles = tiles_x * tiles_y * tiles_z;
    const unsigned n = blockIdx.x / tiles;
    unsigned tb = blockIdx.x - n * tiles;
    const unsigned tzi = tb % tiles_z; tb /= tiles_z;
    const unsigned tyi = tb % tiles_y, txi = tb / tiles_y;
    const int x0 = txi * TX, y0 = tyi * TY, z0 = tzi * TZ;
    const int wx0 = x0 - H, wy0 = y0 - H, wz0 = z0 - ZLO;
    const float *un = u + (size_t)n * 3 * NV, *vn = v + (size_t)n * 3 * NV;
    float *on = out + (size_t)n * 3 * NV;
    const int t = threadIdx.x, lz = t & 31, ly = (t >> 5) & 15, lxb = t >> 9;  // 0 when NT1 == 512
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(un), 0, 3 * NV * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(vn), 0, 3 * NV * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(on, 0, 3 * NV * 4, 0x00020000);

    // window chunk sources of this lane (the same for every channel)
    int csrc[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int c = r * NT1 + t;
        const int row = c / WZC, cz = c - row * WZC;
        const int wx = row / WY, wy = row - wx * WY;
        const int gx = wx0 + wx, gy = wy0 + wy, gz = wz0 + 4 * cz;
        const bool ok = c < NCHUNK && gx >= 0 && gx < NX && gy >= 0 && gy < NY && gz >= 0 && gz < NZ;
        csrc[r] = ok ? (gx * NY + gy) * NZ + gz : -1;
    }
    auto issue = [&](int c, int buf) {
        const float *vc = vn + (size_t)c * NV;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            float *dst = win + buf * WBUF + (r * NT1 + (t & ~63)) * 4;  // wave-uniform; lane l lands at + 16 l
            if (csrc[r] >= 0)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vc + csrc[r]),
                                                 (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
    };
    issue(0, 0);

    float uu[3][U1], tx[U1], ty[U1], tz[U1];
    int wb[U1];  // window index of corner (fx, fy, fz) << 3 | (cx > fx) << 2 | (cy > fy) << 1 | (cz > fz); < 0: global loads
    const unsigned s0 = ((x0 + lxb) * NY + y0 + ly) * NZ + z0 + lz;
#pragma unroll
    for (int e = 0; e < U1; ++e) {
#pragma unroll
        for (int d = 0; d < 3; ++d)
            uu[d][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ru, s0 * 4, (d * NV + e * XS * NY * NZ) * 4, 0));
    }
#pragma unroll
    for (int e = 0; e < U1; ++e) {
        const Pt p = setup(x0 + lxb + XS * e, y0 + ly, z0 + lz, uu[0][e], uu[1][e], uu[2][e]);
        tx[e] = p.tx; ty[e] = p.ty; tz[e] = p.tz;
        const int ax = p.fx - wx0, ay = p.fy - wy0, az = p.fz - wz0;
        const int bx = p.cx - wx0, by = p.cy - wy0, bz = p.cz - wz0;
        const bool in = ax >= 0 && bx < WX && ay >= 0 && by < WY && az >= 0 && bz < WZ;
        wb[e] = in ? ((ax * WY + ay) * WZ + az) << 3 | (p.cx - p.fx) << 2 | (p.cy - p.fy) << 1 | (p.cz - p.fz) : -1;
    }
    if (DB) issue(1, 1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int buf = DB ? (c & 1) : 0;
        // channel c's window has landed once every wave's loads for it are done
        if (DB && c < 2) __builtin_amdgcn_s_waitcnt(0x0f70 | ROUNDS);  // vmcnt <= ROUNDS: only channel c + 1 outstanding
        else __builtin_amdgcn_s_waitcnt(0x0f70);                        // vmcnt(0)
        __syncthreads();
        float o[U1];
#pragma unroll
        for (int e = 0; e < U1; ++e) {
            float c000, c001, c010, c011, c100, c101, c110, c111;
            Pt q;
            q.tx = tx[e]; q.ty = ty[e]; q.tz = tz[e];
            {
                int wi = wb[e] >= 0 ? wb[e] : 0;
                asm volatile("" : "+v"(wi));  // recompute the four addresses per channel instead of keeping them
                const int dx = (wi & 4) ? WY * WZ : 0, dy = (wi & 2) ? WZ : 0, dz = wi & 1;
                const int b = buf * WBUF + (wi >> 3);
                c000 = win[b]; c001 = win[b + 1]; c010 = win[b + dy]; c011 = win[b + dy + 1];
                c100 = win[b + dx]; c101 = win[b + dx + 1]; c110 = win[b + dx + dy]; c111 = win[b + dx + dy + 1];
                c001 = dz ? c001 : c000; c011 = dz ? c011 : c010; c101 = dz ? c101 : c100; c111 = dz ? c111 : c110;
            }
            asm volatile("" : "+v"(c000), "+v"(c001), "+v"(c010), "+v"(c011), "+v"(c100), "+v"(c101), "+v"(c110), "+v"(c111));
            if (FB && __builtin_amdgcn_ballot_w64(wb[e] < 0) != 0 && wb[e] < 0) {
                atomicAdd(nfallback, 1u);
                float a0 = uu[0][e], a1 = uu[1][e], a2 = uu[2][e];
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2));  // recompute here; do not keep six coordinates per voxel alive
                q = setup(x0 + lxb + XS * e, y0 + ly, z0 + lz, a0, a1, a2);
                auto g = [&](int x, int y, int z) {
                    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, ((x * NY + y) * NZ + z) * 4, c * NV * 4, 0));
                };
                c000 = g(q.fx, q.fy, q.fz); c001 = g(q.fx, q.fy, q.cz); c010 = g(q.fx, q.cy, q.fz); c011 = g(q.fx, q.cy, q.cz);
                c100 = g(q.cx, q.fy, q.fz); c101 = g(q.cx, q.fy, q.cz); c110 = g(q.cx, q.cy, q.fz); c111 = g(q.cx, q.cy, q.cz);
            }
            o[e] = ds * uu[c][e] + dt * lerp8(q, c000, c001, c010, c011, c100, c101, c110, c111);
            if (e & 1) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int e = 0; e < U1; ++e)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o[e]), ro, s0 * 4, (c * NV + e * XS * NY * NZ) * 4, 0);
        if (c < 2) {
            if (DB) {
                if (c == 0) { __syncthreads(); issue(2, 0); }  // buffer 0 is free once everyone has read channel 0
            } else {
                __syncthreads();
                issue(c + 1, 0);
            }
        }
    }
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 32;
    const float amp = argc > 2 ? (float)atof(argv[2]) : 1.5f;
    const size_t tot = (size_t)N * 3 * NV;
    std::vector<float> hu(3 * NV), hv(3 * NV);
    for (int i = 0; i < NX; ++i)
        for (int j = 0; j < NY; ++j)
            for (int k = 0; k < NZ; ++k) {
                const size_t ix = ((size_t)i * NY + j) * NZ + k;
                hu[ix] = amp * sinf(0.11f * i + 0.07f * j + 0.05f * k);
                hu[NV + ix] = amp * cosf(0.06f * i - 0.09f * j + 0.08f * k);
                hu[2 * NV + ix] = amp * sinf(0.05f * i + 0.1f * j - 0.12f * k + 1.f);
                for (int c = 0; c < 3; ++c) hv[c * NV + ix] = (float)((ix * 2654435761u + c * 40503u) % 1000) * 1e-3f;
            }
    float *u, *v, *o0, *o1;
    unsigned *nf;
    CK(hipMalloc(&u, tot * 4)); CK(hipMalloc(&v, tot * 4)); CK(hipMalloc(&o0, tot * 4)); CK(hipMalloc(&o1, tot * 4));
    CK(hipMalloc(&nf, 4)); CK(hipMemset(nf, 0, 4));
    for (int n = 0; n < N; ++n) {
        CK(hipMemcpy(u + (size_t)n * 3 * NV, hu.data(), 3 * NV * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(v + (size_t)n * 3 * NV, hv.data(), 3 * NV * 4, hipMemcpyHostToDevice));
    }
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timeit = [&](const char *name, auto launch) {
        for (int i = 0; i < 2; ++i) launch();
        CK(hipEventRecord(a));
        const int reps = 10;
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        printf("%-44s %8.1f us  (%.2f TB/s of the 9 algorithmic planes)\n", name, ms * 1e3 / reps,
               9.0 * N * NV * 4 / (ms * 1e-3 / reps) / 1e12);
    };
    const float ds = 1.f, dt = -0.1f;
    timeit("pair gathers, 256 thr, U=2", [&] {
        hipLaunchKernelGGL(k_pair<2>, dim3((unsigned)(N * NV / 512)), dim3(256), 0, 0, o0, u, v, ds, dt);
    });
    for (int vs = 1; vs >= 0; --vs)
        for (int D : {0, 1, 2, 4, 8, 16, 32}) {
            char name[96];
            snprintf(name, sizeof name, "pair gathers + L2-warming wave, D=%d, v slab +%d", D, vs);
            timeit(name, [&] {
                hipLaunchKernelGGL(k_pair_pf<2>, dim3((unsigned)(N * NV / 512)), dim3(320), 0, 0, o1, u, v, ds, dt, D, vs);
            });
            if (D == 4) {
                std::vector<float> a(3 * NV), b(3 * NV);
                CK(hipMemcpy(a.data(), o0 + (size_t)(N - 1) * 3 * NV, 3 * NV * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(b.data(), o1 + (size_t)(N - 1) * 3 * NV, 3 * NV * 4, hipMemcpyDeviceToHost));
                size_t bad = 0;
                for (size_t i = 0; i < 3 * NV; ++i) bad += a[i] != b[i];
                printf("  %zu values differ from the plain pair form\n", bad);
            }
        }
    const unsigned tiles = (NX / TX) * (NY / TY) * (NZ / TZ);
    std::vector<float> h0(3 * NV), h1(3 * NV);
    auto compare = [&](const char *name) {
        CK(hipMemcpy(h0.data(), o0 + (size_t)(N - 1) * 3 * NV, 3 * NV * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h1.data(), o1 + (size_t)(N - 1) * 3 * NV, 3 * NV * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < 3 * NV; ++i) bad += h0[i] != h1[i];
        unsigned f;
        CK(hipMemcpy(&f, nf, 4, hipMemcpyDeviceToHost));
        printf("  %s: %zu of %zu values differ from the pair form; fallback corner sets so far %u\n", name, bad, 3 * NV, f);
        CK(hipMemset(o1, 0, tot * 4));
    };
    auto win = [&](const char *name, auto kern, int nt, int nbuf) {
        const int rounds = (NCHUNK + nt - 1) / nt, bytes = nbuf * rounds * nt * 16;
        CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        timeit(name, [&] { hipLaunchKernelGGL(kern, dim3(N * tiles), dim3(nt), bytes, 0, o1, u, v, ds, dt, nf); });
        compare(name);
    };
    win("window 1024 thr U=4 wpe4 single fb", k_window<1024, 4, 4, false, true>, 1024, 1);
    win("window 1024 thr U=4 wpe4 single", k_window<1024, 4, 4, false, false>, 1024, 1);
    win("window 1024 thr U=4 wpe8 single", k_window<1024, 4, 8, false, false>, 1024, 1);
    win("window  512 thr U=8 wpe4 single", k_window<512, 8, 4, false, false>, 512, 1);
    win("window  512 thr U=8 wpe6 single", k_window<512, 8, 6, false, false>, 512, 1);
    win("window 1024 thr U=4 wpe4 double", k_window<1024, 4, 4, true, false>, 1024, 2);
    return 0;
}
