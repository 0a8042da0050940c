// Affine interpolation and regridding -- gfx950 HIP kernels.
//
// Replaces cuda/affine.cu of the reference: affine_interp_kernel_{2,3}d
// (:23-112), affine_interp_kernel_backward_{2,3}d (:171-536),
// regrid_forward_kernel_{2,3}d (:612-681), regrid_backward_kernel_{2,3}d
// (:736-800).  Sample positions are analytic (h = A(x - o) + T + o), so only
// the image / grad_out rows stream and the lerp corners are gathers.
//
// The backward pass of affine_interp reduces g (x) (x - o) and g over a whole
// image.  The reference uses one 512-thread block per (n, c) with a shared
// memory tree; here every workgroup reduces its 256 voxels with wavefront
// shuffles (64 lanes), combines its 4 waves through LDS and issues one float
// atomic per matrix entry, so the launch fills all 256 CUs.
#include <stdlib.h>
#include <algorithm>

#include "common.hpp"

#ifndef LAGO_NT_AFFINE_ST
#define LAGO_NT_AFFINE_ST 0
#endif

namespace lago {

// splat.hip
template <typename R>
int affine_splat_lds(R *d_I, const R *go, const R *A, const R *T, int nc, int64_t nn, const Geom &g, bool bc,
                     hipStream_t s, int gate);
template <typename R>
int regrid_splat_lds(R *d_I, const R *go, int64_t nplanes, const Geom &g, const Geom &gs, const double *O,
                     const double *S, hipStream_t s);

template <typename R>
__device__ __forceinline__ R half_extent(int n) {  // `.5*static_cast<Real>(n-1)`, cuda/affine.cu:42-43
    return (R)(.5 * (double)(R)(n - 1));
}

// ------------------------------------------------------------------ affine forward

template <typename R, int DIM, bool BC, int NTS = LAGO_NT_AFFINE_ST>
__global__ __launch_bounds__(kBlock) void affine_fwd_kernel(R *__restrict__ out, const R *__restrict__ I,
                                                            const R *__restrict__ A, const R *__restrict__ T,
                                                            int nc, Geom g) {
    const Vox v = locate(g);
    if (!v.valid) return;
    const size_t nv = g.nvox;
    const R *An = A + (size_t)v.n * DIM * DIM;
    const R *Tn = T + (size_t)v.n * DIM;
    const R *In = BC ? I : I + (size_t)v.n * nc * nv;
    R *on = out + (size_t)v.n * nc * nv + v.s;
    if (DIM == 3) {
        const R ox = half_extent<R>(g.nx), oy = half_extent<R>(g.ny), oz = half_extent<R>(g.nz);
        const R fi = (R)v.i - ox, fj = (R)v.j - oy, fk = (R)v.k - oz;
        const R hx = lg_fma(An[2], fk, lg_fma(An[0], fi, An[1] * fj)) + Tn[0] + ox;
        const R hy = lg_fma(An[5], fk, lg_fma(An[3], fi, An[4] * fj)) + Tn[1] + oy;
        const R hz = lg_fma(An[8], fk, lg_fma(An[6], fi, An[7] * fj)) + Tn[2] + oz;
        Lerp3<R> L;
        L.setup(hx, hy, hz, g.nx, g.ny, g.nz);
        for (int c = 0; c < nc; ++c) st_pol<NTS>(&on[(size_t)c * nv], L.value(In + (size_t)c * nv));
    } else {
        const R ox = half_extent<R>(g.ny), oy = half_extent<R>(g.nz);
        const R fi = (R)v.j - ox, fj = (R)v.k - oy;
        const R hx = lg_fma(An[0], fi, An[1] * fj) + Tn[0] + ox;
        const R hy = lg_fma(An[2], fi, An[3] * fj) + Tn[1] + oy;
        Lerp2<R> L;
        L.setup(hx, hy, g.ny, g.nz);
        for (int c = 0; c < nc; ++c) st_pol<NTS>(&on[(size_t)c * nv], L.value(In + (size_t)c * nv));
    }
}

// ------------------------------------------------------------------ affine backward

template <typename R>
__device__ __forceinline__ R wave_sum(R x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    return x;
}

template <typename R, int DIM, bool BC, bool NEED_I, bool NEED_A, bool NEED_T>
__global__ __launch_bounds__(kBlock) void affine_bwd_kernel(R *__restrict__ d_I, R *__restrict__ d_A,
                                                            R *__restrict__ d_T, const R *__restrict__ go,
                                                            const R *__restrict__ I, const R *__restrict__ A,
                                                            const R *__restrict__ T, int nc, Geom g) {
    constexpr int NP = DIM * DIM + DIM;
    __shared__ R red[kBlock / 64][NP];
    const size_t nv = g.nvox;
    R p[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) p[q] = 0;
    // blockIdx.y = batch item; the x dimension strides over that item's 256-voxel chunks, so the
    // 12 (6) dA/dT atomics per workgroup come from at most gridDim.x workgroups per item (every
    // workgroup adding into the same few addresses is ~14x slower than spread atomics)
    Vox v;
    v.n = blockIdx.y;
    for (uint32_t chunk = blockIdx.x; chunk < g.nbx; chunk += gridDim.x) {
    v.s = chunk * kBlock + threadIdx.x;
    v.valid = v.s < g.nvox;
    {
        const uint32_t sl = v.valid ? v.s : 0;
        const uint32_t ii = g.dyz.div(sl);
        const uint32_t rr = sl - ii * (uint32_t)(g.ny * g.nz);
        const uint32_t jj = g.dz.div(rr);
        v.i = (int)ii; v.j = (int)jj; v.k = (int)(rr - jj * (uint32_t)g.nz);
    }
    if (v.valid) {
        const R *An = A + (size_t)v.n * DIM * DIM;
        const R *Tn = T + (size_t)v.n * DIM;
        const R *In = BC ? I : I + (size_t)v.n * nc * nv;
        R *dIn = NEED_I ? (BC ? d_I : d_I + (size_t)v.n * nc * nv) : nullptr;
        const R *gon = go + (size_t)v.n * nc * nv + v.s;
        if (DIM == 3) {
            const R ox = half_extent<R>(g.nx), oy = half_extent<R>(g.ny), oz = half_extent<R>(g.nz);
            const R fi = (R)v.i - ox, fj = (R)v.j - oy, fk = (R)v.k - oz;
            const R hx = lg_fma(An[2], fk, lg_fma(An[0], fi, An[1] * fj)) + Tn[0] + ox;
            const R hy = lg_fma(An[5], fk, lg_fma(An[3], fi, An[4] * fj)) + Tn[1] + oy;
            const R hz = lg_fma(An[8], fk, lg_fma(An[6], fi, An[7] * fj)) + Tn[2] + oz;
            Splat3<R> S;
            Lerp3<R> L;
            if (NEED_I) S.setup(hx, hy, hz, g.nx, g.ny, g.nz);
            if (NEED_A || NEED_T) L.setup(hx, hy, hz, g.nx, g.ny, g.nz);
            for (int c = 0; c < nc; ++c) {
                const R diff = gon[(size_t)c * nv];
                if (NEED_I) {
                    R *dIc = dIn + (size_t)c * nv;
#pragma unroll
                    for (int q = 0; q < 8; ++q) atomic_add(dIc + S.o[q], S.w[q] * diff);
                }
                if (NEED_A || NEED_T) {
                    R gx, gy, gz;
                    L.grad(In + (size_t)c * nv, gx, gy, gz);
                    gx *= diff; gy *= diff; gz *= diff;  // cuda/affine.cu:415-417
                    if (NEED_A) {
                        p[0] = lg_fma(gx, fi, p[0]); p[1] = lg_fma(gx, fj, p[1]); p[2] = lg_fma(gx, fk, p[2]);
                        p[3] = lg_fma(gy, fi, p[3]); p[4] = lg_fma(gy, fj, p[4]); p[5] = lg_fma(gy, fk, p[5]);
                        p[6] = lg_fma(gz, fi, p[6]); p[7] = lg_fma(gz, fj, p[7]); p[8] = lg_fma(gz, fk, p[8]);
                    }
                    if (NEED_T) { p[9] += gx; p[10] += gy; p[11] += gz; }
                }
            }
        } else {
            const R ox = half_extent<R>(g.ny), oy = half_extent<R>(g.nz);
            const R fi = (R)v.j - ox, fj = (R)v.k - oy;
            const R hx = lg_fma(An[0], fi, An[1] * fj) + Tn[0] + ox;
            const R hy = lg_fma(An[2], fi, An[3] * fj) + Tn[1] + oy;
            Splat2<R> S;
            Lerp2<R> L;
            if (NEED_I) S.setup(hx, hy, g.ny, g.nz);
            if (NEED_A || NEED_T) L.setup(hx, hy, g.ny, g.nz);
            for (int c = 0; c < nc; ++c) {
                const R diff = gon[(size_t)c * nv];
                if (NEED_I) {
                    R *dIc = dIn + (size_t)c * nv;
#pragma unroll
                    for (int q = 0; q < 4; ++q) atomic_add(dIc + S.o[q], S.w[q] * diff);
                }
                if (NEED_A || NEED_T) {
                    R gx, gy;
                    L.grad(In + (size_t)c * nv, gx, gy);
                    gx *= diff; gy *= diff;  // cuda/affine.cu:241-242
                    if (NEED_A) { p[0] = lg_fma(gx, fi, p[0]); p[1] = lg_fma(gx, fj, p[1]); p[2] = lg_fma(gy, fi, p[2]); p[3] = lg_fma(gy, fj, p[3]); }
                    if (NEED_T) { p[4] += gx; p[5] += gy; }
                }
            }
        }
    }
    }  // chunk loop
    if (NEED_A || NEED_T) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            R s = wave_sum(p[q]);
            if (lane == 0) red[wave][q] = s;
        }
        __syncthreads();
        if (threadIdx.x < NP) {
            const int q = threadIdx.x;
            R s = red[0][q] + red[1][q] + red[2][q] + red[3][q];
            if (q < DIM * DIM) {
                if (NEED_A) atomic_add(d_A + (size_t)v.n * DIM * DIM + q, s);
            } else {
                if (NEED_T) atomic_add(d_T + (size_t)v.n * DIM + (q - DIM * DIM), s);
            }
        }
    }
}

std::atomic<int> g_affine_box{1};   // 1: affine_interp_backward's image splat by target boxes where the matrix allows
void tune_affine(int box) { g_affine_box = box ? 1 : 0; }

// ------------------------------------------------------------------ affine backward: image splat by TARGET boxes
//
// The sheared windows of the displacement splat do not carry over to an affine map (a 15 degree rotation moves a tile's
// footprint by four cells across a 16-cell z segment), and the general tiled kernel (splat.hip) covers a rotated tile
// with its bounding box: 270 of the 350 us of affine_interp_backward at 8 x 128^3.  For an affine map the roles can be
// turned round exactly: a workgroup OWNS a box of target cells and processes precisely the source voxels whose
// clamped floor cell lies in it -- a partition of the sources, decided with the reference's own position expression
// (cuda/affine.cu:42-61), so every contribution is made once.  Their eight cells are the box plus one layer on its
// upper faces: the LDS window is (BX + 1)(BY + 1)(BZ + 1) cells, nothing ever misses it, and the flush touches 1.3
// cells per voxel instead of the bounding box of a sheared tile.  Candidates: the source voxels in the bounding box of
// the box's PREIMAGE (the inverse matrix in double, one voxel of slack for the rounding of h; border boxes own
// everything clamped onto them and reach out to the image of the source grid's corners).  Matrices whose inverse would
// make that candidate set large (or that have none) are left, per batch item and decided on the device, to the general
// kernel (common.hpp: affine_item_regular).  d_I as always: float64 sums per window, one float atomic per touched cell.
#ifndef LAGO_BOX_ROWS
#define LAGO_BOX_ROWS 3   // candidate rows a wave of affine_splat_box_kernel keeps in flight (1: 231, 2: 227, 3: 222, 4: 224 us at 8 x 128^3)
#endif
struct BoxGeom {
    int nx, ny, nz, BX, BY, BZ;
    uint32_t nbx, nby, nbz, per_item, total;
    int rev;
    FastDiv d_item, d_yz, d_z, d_wyz, d_wz;
};

template <typename R, bool BC>
__global__ __launch_bounds__(kBlock) void affine_splat_box_kernel(R *__restrict__ d_I, const R *__restrict__ go,
                                                                  const R *__restrict__ A, const R *__restrict__ T, int nc,
                                                                  BoxGeom bg) {
    extern __shared__ __align__(16) unsigned char lago_bx[];
    double *win = reinterpret_cast<double *>(lago_bx);
    const int nx = bg.nx, ny = bg.ny, nz = bg.nz;
    const size_t nv = (size_t)nx * ny * nz;
    const uint32_t L = block_order(blockIdx.x, bg.total, bg.rev);
    const uint32_t n = bg.d_item.div(L);
    uint32_t r = L - n * bg.per_item;
    const uint32_t bx = bg.d_yz.div(r);
    r -= bx * (bg.nby * bg.nbz);
    const uint32_t by = bg.d_z.div(r), bz = r - by * bg.nbz;
    const int X0 = (int)bx * bg.BX, Y0 = (int)by * bg.BY, Z0 = (int)bz * bg.BZ;
    const int ex = min(bg.BX, nx - X0), ey = min(bg.BY, ny - Y0), ez = min(bg.BZ, nz - Z0);
    const R *An = A + (size_t)n * 9, *Tn = T + (size_t)n * 3;
    double Ai[9];
    if (!affine_item_regular<R>(An, Ai)) return;   // this item is the general kernel's
    const R ox = half_extent<R>(nx), oy = half_extent<R>(ny), oz = half_extent<R>(nz);
    const double od[3] = {(double)ox, (double)oy, (double)oz};
    const double Td[3] = {(double)Tn[0], (double)Tn[1], (double)Tn[2]};
    // the box in position space: [lo, hi) per axis; a border box owns everything clamped onto it, i.e. it reaches to
    // the image of the source grid (its eight corners) on that side
    double lo[3] = {(double)X0, (double)Y0, (double)Z0}, hi[3] = {(double)(X0 + ex), (double)(Y0 + ey), (double)(Z0 + ez)};
    {
        double hmin[3] = {1e300, 1e300, 1e300}, hmax[3] = {-1e300, -1e300, -1e300};
        const int ext[3] = {nx, ny, nz};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const double f[3] = {((q & 4) ? ext[0] - 1 : 0) - od[0], ((q & 2) ? ext[1] - 1 : 0) - od[1], ((q & 1) ? ext[2] - 1 : 0) - od[2]};
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const double h = (double)An[3 * d] * f[0] + (double)An[3 * d + 1] * f[1] + (double)An[3 * d + 2] * f[2] + Td[d] + od[d];
                hmin[d] = h < hmin[d] ? h : hmin[d];
                hmax[d] = h > hmax[d] ? h : hmax[d];
            }
        }
        const int org[3] = {X0, Y0, Z0}, len[3] = {ex, ey, ez};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (org[d] == 0) lo[d] = fmin(lo[d], hmin[d] - 2.0);
            if (org[d] + len[d] == ext[d]) hi[d] = fmax(hi[d], hmax[d] + 2.0);
        }
    }
    // candidate sources: bounding box of the preimage of [lo, hi), one voxel of slack, inside the grid
    int s0[3], s1[3];
    {
        double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const double h[3] = {((q & 4) ? hi[0] : lo[0]) - Td[0] - od[0], ((q & 2) ? hi[1] : lo[1]) - Td[1] - od[1],
                                 ((q & 1) ? hi[2] : lo[2]) - Td[2] - od[2]};
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const double x = Ai[3 * d] * h[0] + Ai[3 * d + 1] * h[1] + Ai[3 * d + 2] * h[2] + od[d];
                mn[d] = x < mn[d] ? x : mn[d];
                mx[d] = x > mx[d] ? x : mx[d];
            }
        }
        const int ext[3] = {nx, ny, nz};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            // slack for the float rounding of h and of the inverse, NOT a whole voxel per side: inside the grid |h| is below
            // the extent n, so h is off by at most ~6 float ulps of n = 3.6e-7 n, and the rows of |A^-1| sum to at most 4
            // (affine_item_regular): 1.5e-6 n voxels in source coordinates.  0.02 + 1e-5 n leaves a factor of seven.  A box of
            // 8 x 8 x 48 cells then has about 1.45 instead of 2.0 candidates per owned voxel (-9 % on the kernel,
            // profiles/r06_ab_box_slack.txt); border boxes reach out to the image of the grid's corners as before.
            const double slack = 0.02 + 1e-5 * (double)max(nx, max(ny, nz));
            s0[d] = max(0, (int)floor(fmax(mn[d] - slack, -1e9)));
            s1[d] = min(ext[d] - 1, (int)ceil(fmin(mx[d] + slack, 1e9)));
        }
    }
    const int WY = bg.BY + 1, WZ = bg.BZ + 1;
    const int wcells = (bg.BX + 1) * WY * WZ;
    for (int f = threadIdx.x; f < wcells; f += kBlock) win[f] = 0.0;
    __syncthreads();
    const int cy = s1[1] - s0[1] + 1, cz = s1[2] - s0[2] + 1;
    const int rows = s1[0] >= s0[0] && cy > 0 && cz > 0 ? (s1[0] - s0[0] + 1) * cy : 0;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const R *gon = go + (size_t)n * nc * nv;
    R *dIn = BC ? d_I : d_I + (size_t)n * nc * nv;
    for (int c = 0; c < nc; ++c) {
        const R *gc = gon + (size_t)c * nv;
        // one wave per candidate row (i, j), lanes along z, LAGO_BOX_ROWS rows in flight per wave: every row's positions and
        // ownership first, with its grad_out value requested through a buffer load whose offset is out of range for the
        // lanes that do NOT own their candidate (no memory traffic for them: a plain prefetch of the next row measured
        // 15 % slower in round 4 because it loaded every candidate), then the weights and LDS adds of all of them -- one
        // dependent memory round trip per LAGO_BOX_ROWS rows instead of one per row.
        const BufRsrc rgo = make_rsrc(gc, (uint32_t)(nv * sizeof(R)));
        constexpr int UR = LAGO_BOX_ROWS;
        for (int row0 = wave; row0 < rows; row0 += (kBlock / 64) * UR) {
            for (int kb = s0[2]; kb <= s1[2]; kb += 64) {
                const int k = kb + lane;
                const R fk = (R)k - oz;
                R hx[UR], hy[UR], hz[UR], diff[UR];
                int fx[UR], fy[UR], fz[UR];
                uint32_t lx[UR], ly[UR], lz[UR];
                bool own[UR];
#pragma unroll
                for (int q = 0; q < UR; ++q) {
                    const int row = row0 + q * (kBlock / 64);
                    const int i = s0[0] + row / cy, j = s0[1] + row % cy;   // (scalar)
                    const R fi = (R)i - ox, fj = (R)j - oy;
                    // cuda/affine.cu:42-61 (as affine_bwd_kernel above)
                    hx[q] = lg_fma(An[2], fk, lg_fma(An[0], fi, An[1] * fj)) + Tn[0] + ox;
                    hy[q] = lg_fma(An[5], fk, lg_fma(An[3], fi, An[4] * fj)) + Tn[1] + oy;
                    hz[q] = lg_fma(An[8], fk, lg_fma(An[6], fi, An[7] * fj)) + Tn[2] + oz;
                    fx[q] = lg_floor(hx[q]); fy[q] = lg_floor(hy[q]); fz[q] = lg_floor(hz[q]);
                    lx[q] = (uint32_t)(clamp1(fx[q], nx) - X0); ly[q] = (uint32_t)(clamp1(fy[q], ny) - Y0); lz[q] = (uint32_t)(clamp1(fz[q], nz) - Z0);
                    own[q] = row < rows && k <= s1[2] && lx[q] < (uint32_t)ex && ly[q] < (uint32_t)ey && lz[q] < (uint32_t)ez;   // else: another box owns it
                    const uint32_t off = (uint32_t)((((size_t)i * ny + j) * nz + k) * sizeof(R));
                    diff[q] = buf_load1<R>(rgo, own[q] ? off : 0xffffffffu);
                }
#pragma unroll
                for (int q = 0; q < UR; ++q) {
                    if (!own[q]) continue;
                    const uint32_t lx1 = (uint32_t)(clamp1(fx[q] + 1, nx) - X0), ly1 = (uint32_t)(clamp1(fy[q] + 1, ny) - Y0),
                                   lz1 = (uint32_t)(clamp1(fz[q] + 1, nz) - Z0);
                    const uint32_t cxs[2] = {lx[q], lx1}, cys[2] = {ly[q], ly1}, czs[2] = {lz[q], lz1};
                    // sequentially flipped weights (include/interp.h:431-453): x outer, y, z inner
                    R ddx = (R)1.f - (hx[q] - (R)fx[q]), ddy = (R)1.f - (hy[q] - (R)fy[q]), ddz = (R)1.f - (hz[q] - (R)fz[q]);
#pragma unroll
                    for (int c8 = 0; c8 < 8; ++c8) {
                        const R w = (ddx * ddy * ddz) * diff[q];
                        __hip_atomic_fetch_add(win + ((cxs[c8 >> 2] * (uint32_t)WY + cys[(c8 >> 1) & 1]) * (uint32_t)WZ + czs[c8 & 1]),
                                               (double)w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        ddz = (R)1.f - ddz;
                        if (c8 & 1) ddy = (R)1.f - ddy;
                        if ((c8 & 3) == 3) ddx = (R)1.f - ddx;
                    }
                }
            }
        }
        __syncthreads();
        // flush: one wave per window row (lx, ly), lanes along z; re-zero for the next channel
        R *dIc = dIn + (size_t)c * nv;
        const int fx_n = min(ex + 1, nx - X0), fy_n = min(ey + 1, ny - Y0), fz_n = min(ez + 1, nz - Z0);
        for (int row = wave; row < fx_n * fy_n; row += kBlock / 64) {
            const int lx = row / fy_n, ly = row % fy_n;
            double *wrow = win + (lx * WY + ly) * WZ;
            R *grow = dIc + ((size_t)(X0 + lx) * ny + (Y0 + ly)) * nz + Z0;
            for (int lz = lane; lz < fz_n; lz += 64) {
                const double acc = wrow[lz];
                if (acc != 0.0) {
                    wrow[lz] = 0.0;
                    atomic_add(grow + lz, (R)acc);
                }
            }
        }
        __syncthreads();
    }
}

// Launches the box kernel for the regular batch items and the general kernel, gated, for the others.  Returns 0 when the
// image splat is done, 1 when the shape is left to the plain kernel, < 0 on failure.
template <typename R>
static int affine_splat_boxes(R *d_I, const R *go, const R *A, const R *T, int nc, int64_t nn, const Geom &g, bool bc,
                              hipStream_t s) {
    if (g.nz < 16 || g.nx < 2 || g.ny < 2) return 1;
    BoxGeom bg;
    bg.nx = g.nx; bg.ny = g.ny; bg.nz = g.nz;
    bg.BX = g.nx < 8 ? g.nx : 8;
    bg.BY = g.ny < 8 ? g.ny : 8;
    bg.BZ = g.nz < 48 ? g.nz : 48;
    bg.nbx = (uint32_t)((g.nx + bg.BX - 1) / bg.BX);
    bg.nby = (uint32_t)((g.ny + bg.BY - 1) / bg.BY);
    bg.nbz = (uint32_t)((g.nz + bg.BZ - 1) / bg.BZ);
    bg.per_item = bg.nbx * bg.nby * bg.nbz;
    const int64_t total = (int64_t)bg.per_item * nn;
    if (total <= 0 || total >= (1ll << 31)) return 1;
    bg.total = (uint32_t)total;
    bg.rev = g.rev;
    bg.d_item = FastDiv(bg.per_item);
    bg.d_yz = FastDiv(bg.nby * bg.nbz);
    bg.d_z = FastDiv(bg.nbz);
    const size_t smem = (size_t)(bg.BX + 1) * (bg.BY + 1) * (bg.BZ + 1) * sizeof(double);
    // the items the box kernel leaves: the general tiled kernel (or, where it does not apply, nothing: handled below)
    const int rc = affine_splat_lds<R>(d_I, go, A, T, nc, nn, g, bc, s, 1);
    if (rc != 0) return rc;   // shape not supported by the general kernel either: the caller's plain kernel does all items
    if (bc)
        hipLaunchKernelGGL((affine_splat_box_kernel<R, true>), dim3(bg.total), dim3(kBlock), smem, s, d_I, go, A, T, nc, bg);
    else
        hipLaunchKernelGGL((affine_splat_box_kernel<R, false>), dim3(bg.total), dim3(kBlock), smem, s, d_I, go, A, T, nc, bg);
    note_path(LP_SPLAT_AFFINE_BOX);
    return 0;
}

// ------------------------------------------------------------------ regrid

struct RegridParams {
    double O[3], S[3];
    int nx, ny, nz;  // input (source) extents in geometry order (2D: nx = 1)
};

// Output-grid geometry in g; input extents in rp.
template <typename R, int DIM, int NTS = LAGO_NT_AFFINE_ST>
__global__ __launch_bounds__(kBlock) void regrid_fwd_kernel(R *__restrict__ out, const R *__restrict__ I,
                                                            int nq, Geom g, RegridParams rp) {
    const Vox v = locate(g);  // v.n unused: (n, c) planes are looped here
    if (!v.valid) return;
    const size_t Nv = g.nvox;
    const size_t nvin = (size_t)rp.nx * rp.ny * rp.nz;
    if (DIM == 3) {
        const R Ox = (R)rp.O[0], Oy = (R)rp.O[1], Oz = (R)rp.O[2];
        const R Sx = (R)rp.S[0], Sy = (R)rp.S[1], Sz = (R)rp.S[2];
        const R ox = half_extent<R>(g.nx), oy = half_extent<R>(g.ny), oz = half_extent<R>(g.nz);
        const R hx = lg_fma((R)v.i - ox, Sx, Ox);
        const R hy = lg_fma((R)v.j - oy, Sy, Oy);
        // The reference advances hz by `hz += Sz` per output k (cuda/affine.cu:669-675):
        // a sequentially rounded running sum, reproduced here so positions match bit for bit.
        R hz = lg_fma(-oz, Sz, Oz);
        for (int k = 0; k < v.k; ++k) hz += Sz;
        Lerp3<R> L;
        L.setup(hx, hy, hz, rp.nx, rp.ny, rp.nz);
        for (int q = 0; q < nq; ++q) st_pol<NTS>(&out[(size_t)q * Nv + v.s], L.value(I + (size_t)q * nvin));
    } else {
        const R Ox = (R)rp.O[0], Oy = (R)rp.O[1];
        const R Sx = (R)rp.S[0], Sy = (R)rp.S[1];
        const R ox = half_extent<R>(g.ny), oy = half_extent<R>(g.nz);
        const R hx = lg_fma((R)v.j - ox, Sx, Ox);
        const R hy = lg_fma((R)v.k - oy, Sy, Oy);
        Lerp2<R> L;
        L.setup(hx, hy, rp.ny, rp.nz);
        for (int q = 0; q < nq; ++q) st_pol<NTS>(&out[(size_t)q * Nv + v.s], L.value(I + (size_t)q * nvin));
    }
}

template <typename R, int DIM>
__global__ __launch_bounds__(kBlock) void regrid_bwd_kernel(R *__restrict__ d_I, const R *__restrict__ go, int nq,
                                                            Geom g, RegridParams rp) {
    const Vox v = locate(g);
    if (!v.valid) return;
    const size_t Nv = g.nvox;
    const size_t nvin = (size_t)rp.nx * rp.ny * rp.nz;
    if (DIM == 3) {
        const R Ox = (R)rp.O[0], Oy = (R)rp.O[1], Oz = (R)rp.O[2];
        const R Sx = (R)rp.S[0], Sy = (R)rp.S[1], Sz = (R)rp.S[2];
        const R ox = half_extent<R>(g.nx), oy = half_extent<R>(g.ny), oz = half_extent<R>(g.nz);
        const R hx = lg_fma((R)v.i - ox, Sx, Ox);
        const R hy = lg_fma((R)v.j - oy, Sy, Oy);
        const R hz = lg_fma((R)v.k - oz, Sz, Oz);  // cuda/affine.cu:791: per voxel, no running sum
        Splat3<R> S;
        S.setup(hx, hy, hz, rp.nx, rp.ny, rp.nz);
        for (int q = 0; q < nq; ++q) {
            const R m = go[(size_t)q * Nv + v.s];
            R *d = d_I + (size_t)q * nvin;
#pragma unroll
            for (int e = 0; e < 8; ++e) atomic_add(d + S.o[e], S.w[e] * m);
        }
    } else {
        const R Ox = (R)rp.O[0], Oy = (R)rp.O[1];
        const R Sx = (R)rp.S[0], Sy = (R)rp.S[1];
        const R ox = half_extent<R>(g.ny), oy = half_extent<R>(g.nz);
        const R hx = lg_fma((R)v.j - ox, Sx, Ox);
        const R hy = lg_fma((R)v.k - oy, Sy, Oy);
        Splat2<R> S;
        S.setup(hx, hy, rp.ny, rp.nz);
        for (int q = 0; q < nq; ++q) {
            const R m = go[(size_t)q * Nv + v.s];
            R *d = d_I + (size_t)q * nvin;
#pragma unroll
            for (int e = 0; e < 4; ++e) atomic_add(d + S.o[e], S.w[e] * m);
        }
    }
}

// ------------------------------------------------------------------ regrid backward, separable
//
// regrid_backward_kernel_3d (cuda/affine.cu:767-855) splats every output voxel's value onto eight input cells with the
// product weights wx wy wz of an AXIS-ALIGNED map h = (X - C) S + O: the operator is the transpose of Rx (x) Ry (x) Rz,
// one 1D linear interpolation per axis, and clamping is per axis too.  Applied axis by axis in GATHER form it needs no
// atomics, no zero-initialised target and no float64 LDS window: for input cell z along an axis, the output positions
// that contribute form one index range (h is monotone for S > 0), each tested EXACTLY -- the same position
// expression, floor, clamp and sequentially flipped weight pair as the splat -- so only the association of the
// products differs from the reference (wx wy wz m against wz (wy (wx m))): within the bound that its unordered atomic
// sums leave anyway.  Three passes, each shrinking (upsampling's backward) or growing one axis; the two temporaries come
// from the caller (no allocation inside the library).  Results do not depend on the launch (no atomics).
struct SepAxis {
    int Nout, nin;        // extent of this axis in grad_out's grid / in d_I's grid
    uint32_t Q;           // elements of the faster axes (contiguous run per index of this axis)
    uint32_t total;       // P * nin * Q threads
    double O, S, o;       // origin, spacing, half extent of the OUTPUT grid along this axis (as the splat takes them)
    int slack;            // candidate window: the double inverse map +- slack output indices (sep_axis_slack)
    FastDiv dQ, dn;
};

template <typename R>
__global__ __launch_bounds__(kBlock) void regrid_bwd_axis_kernel(R *__restrict__ T, const R *__restrict__ G, SepAxis ax) {
    const uint32_t idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= ax.total) return;
    const uint32_t pz = ax.dQ.div(idx), q = idx - pz * ax.Q;
    const uint32_t p = ax.dn.div(pz);
    const int zi = (int)(pz - p * (uint32_t)ax.nin);
    const R O = (R)ax.O, S = (R)ax.S, o = (R)ax.o;   // (half_extent<R>: already rounded through R by the host)
    // a superset of the contributing output indices from the inverse map in double, +- ax.slack for the rounding of h
    // in R (the host bounds it: sep_axis_slack); clamped in double, so the casts cannot overflow; the border cells
    // collect everything that is clamped onto them
    int lo = 0, hi = ax.Nout - 1;
    if (zi > 0) lo = (int)fmax(0.0, fmin((double)(ax.Nout - 1), floor(((double)zi - 1.0 - ax.O) / ax.S + ax.o) - (double)ax.slack));
    if (zi < ax.nin - 1) hi = (int)fmax(0.0, fmin((double)(ax.Nout - 1), ceil(((double)zi + 1.0 - ax.O) / ax.S + ax.o) + (double)ax.slack));
    const R *g = G + ((size_t)p * ax.Nout) * ax.Q + q;
    R acc = (R)0;
    // branch-free body, four candidates per trip: the loads do not wait for the membership tests (a candidate that does
    // not contribute is added with weight 0: exact)
    for (int z0 = lo; z0 <= hi; z0 += 4) {
        R v[4], w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int zo = min(z0 + e, hi);
            v[e] = g[(size_t)zo * ax.Q];
            const R h = lg_fma((R)zo - o, S, O);          // cuda/affine.cu:791
            const int f = lg_floor(h);
            const int c0 = clamp1(f, ax.nin), c1 = clamp1(f + 1, ax.nin);
            const R w0 = (R)1.f - (h - (R)f), w1 = (R)1.f - w0;   // include/interp.h:431-453: dz, then 1 - dz
            // both corners on this cell (clamped borders, extent 1): the reference adds them one after the other
            w[e] = z0 + e <= hi ? ((c0 == zi ? w0 : (R)0) + (c1 == zi ? w1 : (R)0)) : (R)0;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = lg_fma(w[e], v[e], acc);
    }
    T[idx] = acc;
}

// The same pass for an axis that is NOT the contiguous one (Q >= 64): one thread per (p, q) line streams along the axis --
// every value of G is loaded once, coalesced over q; the position, floor, clamps and the weight pair depend on the
// output index only, so they are wave-uniform; the cell indices never decrease (S > 0), so two running sums (cells
// `cur` and `cur + 1`) suffice and a finished cell is stored the moment the floor moves past it.  Contributions reach a
// cell in ascending output index: the sums are reproducible.
template <typename R>
__global__ __launch_bounds__(kBlock) void regrid_bwd_axis_stream_kernel(R *__restrict__ T, const R *__restrict__ G, SepAxis ax,
                                                                        uint32_t lines) {
    const uint32_t idx = blockIdx.x * kBlock + threadIdx.x;   // (p, q)
    if (idx >= lines) return;
    const uint32_t p = ax.dQ.div(idx), q = idx - p * ax.Q;
    const R O = (R)ax.O, S = (R)ax.S, o = (R)ax.o;
    const R *g = G + ((size_t)p * ax.Nout) * ax.Q + q;
    R *t = T + ((size_t)p * ax.nin) * ax.Q + q;
    int cur = 0;            // lowest cell not yet stored
    R a0 = (R)0, a1 = (R)0; // sums of cells cur and cur + 1
    for (int zo = 0; zo < ax.Nout; ++zo) {
        const R v = g[(size_t)zo * ax.Q];
        const R h = lg_fma((R)zo - o, S, O);          // cuda/affine.cu:791   (wave-uniform from here ...)
        const int f = lg_floor(h);
        const int c0 = clamp1(f, ax.nin), c1 = clamp1(f + 1, ax.nin);
        const R w0 = (R)1.f - (h - (R)f), w1 = (R)1.f - w0;   // include/interp.h:431-453: dz, then 1 - dz
        while (cur < c0) {                              // (... to here: the loop is scalar)
            t[(size_t)cur * ax.Q] = a0;
            a0 = a1;
            a1 = (R)0;
            ++cur;
        }
        a0 = lg_fma(w0, v, a0);                         // c0 == cur
        if (c1 == cur) a0 = lg_fma(w1, v, a0);
        else a1 = lg_fma(w1, v, a1);                    // c1 == cur + 1
    }
    for (; cur < ax.nin; ++cur) {
        t[(size_t)cur * ax.Q] = a0;
        a0 = a1;
        a1 = (R)0;
    }
}

// How far (in output indices) the position h = fma(zo - o, S, O) evaluated in R can sit from its exact value: the three
// roundings (zo - o, the fma, o itself) are each within eps |h|-sized terms, |h| <= |O| + Nout S; divided by S that is
// an index distance.  The gather kernel widens its candidate window by 1 + that (with a tiny spacing the
// fixed +- 1 did not cover the rounding).
template <typename R>
static double sep_axis_slack(double O, double S, int64_t Nout) {
    const double eps = sizeof(R) == 4 ? 1.1920928955078125e-7 : 2.220446049250313e-16;
    return 1.0 + ceil(4.0 * eps * (fabs(O) + (double)Nout * S + 1.0) / S);
}

template <typename R>
static int regrid_backward_impl(R *d_I, const R *go, int dim, int64_t nn, int64_t nc, int64_t nx, int64_t ny,
                                int64_t nz, int64_t Nx, int64_t Ny, int64_t Nz, const double *origin,
                                const double *spacing, void *stream);

// Separable where the map allows it; every input the reference's splat accepts is accepted here too: what the passes do
// not cover (non-positive, huge or tiny spacings, far origins, 2^31 elements or more per pass) goes to
// regrid_backward_impl, which needs no workspace.
template <typename R>
static int regrid_backward_sep_impl(R *d_I, const R *go, R *ws, int64_t ws_elems, int dim, int64_t nn, int64_t nc,
                                    int64_t nx, int64_t ny, int64_t nz, int64_t Nx, int64_t Ny, int64_t Nz,
                                    const double *origin, const double *spacing, void *stream) {
    if (dim != 2 && dim != 3) return fail_invalid("Only two- and three-dimensional regridding is supported");
    if (!origin || !spacing) return fail_invalid("regrid_backward: bad extent");
    const int64_t a_nx = nx, a_ny = ny, a_nz = nz, a_Nx = Nx, a_Ny = Ny, a_Nz = Nz;   // as the caller gave them
    if (dim == 2) { nz = ny; ny = nx; nx = 1; Nz = Ny; Ny = Nx; Nx = 1; }
    const int64_t planes = nn * nc;
    if (nn < 0 || nc < 0 || nx < 1 || ny < 1 || nz < 1 || Nx < 1 || Ny < 1 || Nz < 1) return fail_invalid("regrid_backward: bad extent");
    const int64_t n_in[3] = {nx, ny, nz}, N_out[3] = {Nx, Ny, Nz};
    double O[3] = {0, 0, 0}, S[3] = {1, 1, 1};
    for (int d = 0; d < dim; ++d) { O[3 - dim + d] = origin[d]; S[3 - dim + d] = spacing[d]; }
    const int first = 3 - dim;
    int slack[3] = {1, 1, 1};
    bool sep = true;
    for (int d = first; d < 3; ++d) {
        if (!(S[d] > 0.0) || !(S[d] < 1e6) || !(fabs(O[d]) < 1e9)) { sep = false; break; }
        const double sl = sep_axis_slack<R>(O[d], S[d], N_out[d]);
        if (!(sl <= 64.0)) { sep = false; break; }
        slack[d] = (int)sl;
    }
    {   // element counts of every pass: 32-bit indices with the multiply-high division (n < 2^31, common.hpp: FastDiv)
        int64_t cur[3] = {Nx, Ny, Nz};
        for (int a = first; a < 3 && sep; ++a) {
            if (planes * cur[0] * cur[1] * cur[2] >= (1ll << 40)) sep = false;
            cur[a] = n_in[a];
            if (planes * cur[0] * cur[1] * cur[2] >= (1ll << 31)) sep = false;
        }
    }
    if (!sep) return regrid_backward_impl<R>(d_I, go, dim, nn, nc, a_nx, a_ny, a_nz, a_Nx, a_Ny, a_Nz, origin, spacing, stream);
    hipStream_t s = (hipStream_t)stream;
    if (planes == 0 || nx * ny * nz == 0) return LAGO_OK;
    // passes over the axes x, y, z (2D: y, z), slowest first: extents (cx, cy, cz) go from the output grid to the input grid
    int64_t cur[3] = {Nx, Ny, Nz};
    const R *src = go;
    for (int a = first; a < 3; ++a) {
        int64_t nxt[3] = {cur[0], cur[1], cur[2]};
        nxt[a] = n_in[a];
        const int64_t out_elems = planes * nxt[0] * nxt[1] * nxt[2];
        const bool last = a == 2;
        // temporaries alternate between the two halves of the workspace
        R *dst = last ? d_I : ws + ((a - first) & 1 ? ws_elems / 2 : 0);
        if (!last && (out_elems > ws_elems / 2 || !ws)) return fail_invalid("regrid_backward (separable): workspace too small");
        SepAxis ax;
        ax.Nout = (int)N_out[a];
        ax.nin = (int)n_in[a];
        int64_t Q = 1;
        for (int d = a + 1; d < 3; ++d) Q *= cur[d];
        ax.Q = (uint32_t)Q;
        ax.total = (uint32_t)out_elems;
        ax.O = O[a];
        ax.S = S[a];
        ax.slack = slack[a];
        ax.o = (double)(R)(.5 * (double)(R)((int)N_out[a] - 1));   // half_extent<R> (device helper), on the host
        ax.dQ = FastDiv(ax.Q);
        ax.dn = FastDiv((uint32_t)ax.nin);
        if (!dst || !src) return fail_invalid("regrid_backward: null pointer");
        const int64_t lines = out_elems / ax.nin;   // (p, q) pairs
        // the contiguous axis (Q == 1) takes the gather form: an LDS-transposed streaming kernel measured 5 % faster
        // for 80^3 -> 160^3 and 4-25 % slower for 128^3 -> 64^3 / 128^3, a strided streaming one 1.3-2.9 x slower
        if (Q >= 64 && lines >= 4096)
            hipLaunchKernelGGL((regrid_bwd_axis_stream_kernel<R>), dim3((uint32_t)((lines + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, dst, src, ax, (uint32_t)lines);
        else
            hipLaunchKernelGGL((regrid_bwd_axis_kernel<R>), dim3((uint32_t)((out_elems + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, dst, src, ax);
        src = dst;
        cur[a] = n_in[a];
    }
    return finish_launch(s, "regrid_backward");
}

// ------------------------------------------------------------------ host entry points

template <typename R>
static int affine_forward_impl(R *out, const R *I, const R *A, const R *T, int dim, int64_t nn, int64_t nc,
                               int64_t nx, int64_t ny, int64_t nz, int bc, void *stream) {
    if (dim != 2 && dim != 3)
        return fail_invalid("Only two- and three-dimensional affine interpolation is supported");
    Geom g;
    if (nc < 0 || !make_geom(g, dim, nn, nx, ny, nz)) return fail_invalid("affine_interp_forward: bad extent");
    if (g.nblocks == 0 || nc == 0) return LAGO_OK;
    if (!out || !I || !A || !T) return fail_invalid("affine_interp_forward: null pointer");
    hipStream_t s = (hipStream_t)stream;
    // Output stores non-temporal once input + output exceed the Infinity Cache (8 x 3 x 128^3: 209 -> 140 us; C = 1,
    // 134 MB in all: no difference).  The kernel is bound by the rate of its four pair gathers per output value, like
    // every gather kernel of the library (profiles/r05_affine_regrid_forward.md): several voxels per lane changed nothing.
    const bool big = (double)(nn + (bc ? 1 : nn)) * nc * g.nvox * sizeof(R) > 256.0 * 1024 * 1024;
#define LAUNCH(D, B)                                                                                                     \
    do {                                                                                                                 \
        if (big) hipLaunchKernelGGL((affine_fwd_kernel<R, D, B, 1>), dim3(g.nblocks), dim3(kBlock), 0, s, out, I, A, T, (int)nc, g); \
        else hipLaunchKernelGGL((affine_fwd_kernel<R, D, B, 0>), dim3(g.nblocks), dim3(kBlock), 0, s, out, I, A, T, (int)nc, g);     \
    } while (0)
    if (dim == 3) {
        if (bc) LAUNCH(3, true); else LAUNCH(3, false);
    } else {
        if (bc) LAUNCH(2, true); else LAUNCH(2, false);
    }
#undef LAUNCH
    return finish_launch(s, "affine_interp_forward");
}

template <typename R, int DIM, bool BC>
static void launch_affine_bwd(R *d_I, R *d_A, R *d_T, const R *go, const R *I, const R *A, const R *T, int nc,
                              const Geom &g, int64_t nn, bool nI, bool nA, bool nT, hipStream_t s) {
    // ~4096 workgroups in total, at most one per 256-voxel chunk
    uint32_t gx = (uint32_t)((4096 + nn - 1) / nn);
    if (gx > g.nbx) gx = g.nbx;
    if (gx < 1) gx = 1;
#define LAUNCH(a, b, c)                                                                                             \
    hipLaunchKernelGGL((affine_bwd_kernel<R, DIM, BC, a, b, c>), dim3(gx, (uint32_t)nn), dim3(kBlock), 0, s, d_I, d_A,  \
                       d_T, go, I, A, T, nc, g)
    const int m = (nI ? 4 : 0) | (nA ? 2 : 0) | (nT ? 1 : 0);
    switch (m) {
        case 7: LAUNCH(true, true, true); break;
        case 6: LAUNCH(true, true, false); break;
        case 5: LAUNCH(true, false, true); break;
        case 4: LAUNCH(true, false, false); break;
        case 3: LAUNCH(false, true, true); break;
        case 2: LAUNCH(false, true, false); break;
        case 1: LAUNCH(false, false, true); break;
        default: break;
    }
#undef LAUNCH
}

template <typename R>
static int affine_backward_impl(R *d_I, R *d_A, R *d_T, const R *go, const R *I, const R *A, const R *T, int dim,
                                int64_t nn, int64_t nc, int64_t nx, int64_t ny, int64_t nz, int bc, int need_I,
                                int need_A, int need_T, void *stream) {
    if (dim != 2 && dim != 3)
        return fail_invalid("Only two- and three-dimensional affine interpolation is supported");
    Geom g;
    if (nc < 0 || !make_geom(g, dim, nn, nx, ny, nz, kBlock, true)) return fail_invalid("affine_interp_backward: bad extent");
    hipStream_t s = (hipStream_t)stream;
    const size_t nI = (size_t)(bc ? 1 : nn) * nc * g.nvox;
    if ((need_I && nI && !d_I) || (need_A && nn && !d_A) || (need_T && nn && !d_T) ||
        (g.nblocks && nc && (!go || !I || !A || !T)))
        return fail_invalid("affine_interp_backward: null pointer");
    if (need_I && nI) LAGO_HIP_TRY(hipMemsetAsync(d_I, 0, nI * sizeof(R), s));
    if (need_A && nn) LAGO_HIP_TRY(hipMemsetAsync(d_A, 0, (size_t)nn * dim * dim * sizeof(R), s));
    if (need_T && nn) LAGO_HIP_TRY(hipMemsetAsync(d_T, 0, (size_t)nn * dim * sizeof(R), s));
    if (nn > 65535) return fail_invalid("affine_interp_backward: batch size above 65535 is not supported");
    if (g.nblocks && nc && need_I && dim == 3 && g_splat_mode >= 1) {
        // the image splat goes through the LDS-privatised kernel; d_A / d_T keep the reduction kernel
        // regular matrices: by target boxes; the others (decided per item on the device): the general tiled kernel
        int rc = g_affine_box ? affine_splat_boxes<R>(d_I, go, A, T, (int)nc, nn, g, bc != 0, s) : 1;
        if (rc == 1) rc = affine_splat_lds<R>(d_I, go, A, T, (int)nc, nn, g, bc != 0, s, 0);
        if (rc < 0) return rc;
        if (rc == 0) need_I = 0;
    }
    if (g.nblocks && nc && (need_I || need_A || need_T)) {
        if (dim == 3) {
            if (bc) launch_affine_bwd<R, 3, true>(d_I, d_A, d_T, go, I, A, T, (int)nc, g, nn, need_I, need_A, need_T, s);
            else launch_affine_bwd<R, 3, false>(d_I, d_A, d_T, go, I, A, T, (int)nc, g, nn, need_I, need_A, need_T, s);
        } else {
            if (bc) launch_affine_bwd<R, 2, true>(d_I, d_A, d_T, go, I, A, T, (int)nc, g, nn, need_I, need_A, need_T, s);
            else launch_affine_bwd<R, 2, false>(d_I, d_A, d_T, go, I, A, T, (int)nc, g, nn, need_I, need_A, need_T, s);
        }
    }
    return finish_launch(s, "affine_interp_backward");
}

static bool make_regrid(RegridParams &rp, int dim, int64_t nx, int64_t ny, int64_t nz, const double *origin,
                        const double *spacing) {
    if (!origin || !spacing) return false;
    if (dim == 2) {
        nz = ny;
        ny = nx;
        nx = 1;
    }
    if (nx < 1 || ny < 1 || nz < 1 || nx * ny * nz >= (1ll << 31)) return false;
    rp.nx = (int)nx;
    rp.ny = (int)ny;
    rp.nz = (int)nz;
    for (int d = 0; d < 3; ++d) {
        rp.O[d] = d < dim ? origin[d] : 0.0;
        rp.S[d] = d < dim ? spacing[d] : 0.0;
    }
    return true;
}

template <typename R>
static int regrid_forward_impl(R *out, const R *I, int dim, int64_t nn, int64_t nc, int64_t nx, int64_t ny,
                               int64_t nz, int64_t Nx, int64_t Ny, int64_t Nz, const double *origin,
                               const double *spacing, void *stream) {
    if (dim != 2 && dim != 3) return fail_invalid("Only two- and three-dimensional regridding is supported");
    Geom g;
    RegridParams rp;
    if (nn < 0 || nc < 0 || nn * nc >= (1ll << 31) || !make_geom(g, dim, 1, Nx, Ny, Nz) ||
        !make_regrid(rp, dim, nx, ny, nz, origin, spacing))
        return fail_invalid("regrid_forward: bad extent");
    if (g.nblocks == 0 || nn * nc == 0) return LAGO_OK;
    if (!out || !I) return fail_invalid("regrid_forward: null pointer");
    hipStream_t s = (hipStream_t)stream;
    // (non-temporal stores once input + output exceed the Infinity Cache: 80^3 -> 160^3 at 8 x 3 planes 260 -> 190 us)
    const bool big = ((double)nn * nc * ((double)g.nvox + (double)rp.nx * rp.ny * rp.nz)) * sizeof(R) > 256.0 * 1024 * 1024;
#define LAUNCH(D)                                                                                                        \
    do {                                                                                                                 \
        if (big) hipLaunchKernelGGL((regrid_fwd_kernel<R, D, 1>), dim3(g.nblocks), dim3(kBlock), 0, s, out, I, (int)(nn * nc), g, rp); \
        else hipLaunchKernelGGL((regrid_fwd_kernel<R, D, 0>), dim3(g.nblocks), dim3(kBlock), 0, s, out, I, (int)(nn * nc), g, rp);     \
    } while (0)
    if (dim == 3) LAUNCH(3); else LAUNCH(2);
#undef LAUNCH
    return finish_launch(s, "regrid_forward");
}

template <typename R>
static int regrid_backward_impl(R *d_I, const R *go, int dim, int64_t nn, int64_t nc, int64_t nx, int64_t ny,
                                int64_t nz, int64_t Nx, int64_t Ny, int64_t Nz, const double *origin,
                                const double *spacing, void *stream) {
    if (dim != 2 && dim != 3) return fail_invalid("Only two- and three-dimensional regridding is supported");
    Geom g;
    RegridParams rp;
    if (nn < 0 || nc < 0 || nn * nc >= (1ll << 31) || !make_geom(g, dim, 1, Nx, Ny, Nz, kBlock, true) ||
        !make_regrid(rp, dim, nx, ny, nz, origin, spacing))
        return fail_invalid("regrid_backward: bad extent");
    hipStream_t s = (hipStream_t)stream;
    const size_t nI = (size_t)nn * nc * rp.nx * rp.ny * rp.nz;
    if ((nI && !d_I) || (nI && g.nblocks && !go)) return fail_invalid("regrid_backward: null pointer");
    if (nI) LAGO_HIP_TRY(hipMemsetAsync(d_I, 0, nI * sizeof(R), s));
    if (g.nblocks && nn * nc) {
        if (dim == 3 && g_splat_mode >= 1) {
            Geom gt;
            if (make_geom(gt, 3, 1, nx, ny, nz, kBlock, true)) {
                const int rc = regrid_splat_lds<R>(d_I, go, nn * nc, gt, g, rp.O, rp.S, s);
                if (rc <= 0) return rc;  // done (or failed); 1 = shape left to the plain kernel
            }
        }
        note_path(LP_SPLAT_GLOBAL);
        if (dim == 3)
            hipLaunchKernelGGL((regrid_bwd_kernel<R, 3>), dim3(g.nblocks), dim3(kBlock), 0, s, d_I, go, (int)(nn * nc), g, rp);
        else
            hipLaunchKernelGGL((regrid_bwd_kernel<R, 2>), dim3(g.nblocks), dim3(kBlock), 0, s, d_I, go, (int)(nn * nc), g, rp);
    }
    return finish_launch(s, "regrid_backward");
}

}  // namespace lago

extern "C" {
#define LAGO_DEFINE(REAL, SUF)                                                                                      \
    int lago_affine_interp_forward##SUF(REAL *out, const REAL *I, const REAL *A, const REAL *T, int dim,           \
                                        int64_t nn, int64_t nc, int64_t nx, int64_t ny, int64_t nz, int bc,        \
                                        void *stream) {                                                            \
        return lago::affine_forward_impl<REAL>(out, I, A, T, dim, nn, nc, nx, ny, nz, bc, stream);                 \
    }                                                                                                               \
    int lago_affine_interp_backward##SUF(REAL *d_I, REAL *d_A, REAL *d_T, const REAL *go, const REAL *I,           \
                                         const REAL *A, const REAL *T, int dim, int64_t nn, int64_t nc,            \
                                         int64_t nx, int64_t ny, int64_t nz, int bc, int need_I, int need_A,       \
                                         int need_T, void *stream) {                                               \
        return lago::affine_backward_impl<REAL>(d_I, d_A, d_T, go, I, A, T, dim, nn, nc, nx, ny, nz, bc, need_I,   \
                                                need_A, need_T, stream);                                           \
    }                                                                                                               \
    int lago_regrid_forward##SUF(REAL *out, const REAL *I, int dim, int64_t nn, int64_t nc, int64_t nx,            \
                                 int64_t ny, int64_t nz, int64_t Nx, int64_t Ny, int64_t Nz, const double *origin, \
                                 const double *spacing, void *stream) {                                            \
        return lago::regrid_forward_impl<REAL>(out, I, dim, nn, nc, nx, ny, nz, Nx, Ny, Nz, origin, spacing,       \
                                               stream);                                                            \
    }                                                                                                               \
    int lago_regrid_backward_sep##SUF(REAL *d_I, const REAL *go, REAL *ws, int64_t ws_elems, int dim, int64_t nn,   \
                                      int64_t nc, int64_t nx, int64_t ny, int64_t nz, int64_t Nx, int64_t Ny,       \
                                      int64_t Nz, const double *origin, const double *spacing, void *stream) {      \
        return lago::regrid_backward_sep_impl<REAL>(d_I, go, ws, ws_elems, dim, nn, nc, nx, ny, nz, Nx, Ny, Nz,     \
                                                    origin, spacing, stream);                                       \
    }                                                                                                              \
    int lago_regrid_backward##SUF(REAL *d_I, const REAL *go, int dim, int64_t nn, int64_t nc, int64_t nx,          \
                                  int64_t ny, int64_t nz, int64_t Nx, int64_t Ny, int64_t Nz,                      \
                                  const double *origin, const double *spacing, void *stream) {                     \
        return lago::regrid_backward_impl<REAL>(d_I, go, dim, nn, nc, nx, ny, nz, Nx, Ny, Nz, origin, spacing,     \
                                                stream);                                                           \
    }
LAGO_DEFINE(float, _f32)
LAGO_DEFINE(double, _f64)
#undef LAGO_DEFINE
}
