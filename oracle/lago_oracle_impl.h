/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the lagomorph hot path.
 *
 * Plain scalar C restatement of the reference's CUDA kernels, evaluated under
 * strict IEEE semantics (compile with -ffp-contract=off).  Included twice by
 * lago_oracle.c, once with REAL=float and once with REAL=double.  Nothing under
 * lagomorph_amd/ may link, import or call this file; it exists so that tests,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg have a checker.
 *
 * Every function cites the reference file:line (relative to
 * /root/reference/lagomorph/extension/) whose arithmetic it follows.  Loop
 * *order* over voxels is free (the reference is a parallel kernel); the order
 * of floating point operations inside one voxel is the reference's.
 *
 * Contraction: every `a*b + c` of the reference is written LG_FMA(a, b, c), and a
 * sum of two products `a*b + c*d` as LG_FMA(a, b, c*d) (left product fused), the
 * pattern nvcc's default -fmad=true / LLVM's contraction produces.  lago_oracle.c
 * is built twice: with LG_FMA = fused multiply-add (liblago_oracle.so, bit-
 * comparable with the HIP kernels, which use the identical pattern) and with
 * LG_FMA(a,b,c) = a*b + c, strictly unfused (liblago_oracle_strict.so, bit-
 * comparable with oracle/_ref, the reference's CPU source built with
 * -ffp-contract=off).
 */

#define LG_CAT_(a, b) a##b
#define LG_CAT(a, b) LG_CAT_(a, b)
#define FN(name) LG_CAT(name, SUF)

/* ---- index helpers -------------------------------------------------------- */

/* include/interp.h:64-70 -- truncation toward zero, then step down for
 * negative non-integers.  Positions are saturated to +-2^30 first so that the
 * int conversion is defined for every finite input (the reference has UB
 * there); any such position is clamped to the border anyway. */
static inline int FN(lg_floor)(REAL x) {
    if (x > (REAL)1073741824.0) x = (REAL)1073741824.0;
    if (x < (REAL)-1073741824.0) x = (REAL)-1073741824.0;
    int f = (int)x;
    if (x < 0 && x != (REAL)f) --f;
    return f;
}

/* include/extrap.h:46-57 clampBackground(floor, ceil, size).  `size` is a size_t there, so the two upper comparisons are
 * UNSIGNED: a negative ceil beside a non-negative floor -- not an interpolation footprint (ceil = floor + 1), no kernel
 * produces it -- is sent to size - 1.  Restated as written (tests/test_oracle_ref.py pins it against the header). */
static inline void FN(lg_clamp_pair)(int *fl, int *ce, long size) {
    if (*fl < 0) {
        *fl = 0;
        if (*ce < 0) *ce = 0;
    }
    if ((unsigned long)(long)*ce >= (unsigned long)size) {
        *ce = (int)(size - 1);
        if ((unsigned long)(long)*fl >= (unsigned long)size) *fl = (int)(size - 1);
    }
}

/* include/extrap.h:41-44 clamp(r, b) */
static inline int FN(lg_clamp)(int r, long b) {
    if (r < 0) return 0;
    if (r >= b) return (int)(b - 1);
    return r;
}

/* ---- bilinear / trilinear gather ------------------------------------------ */

/* include/interp.h:10-56 biLerp (clamp strategy, the only one instantiated) */
static REAL FN(lg_bilerp)(const REAL *img, REAL x, REAL y, long sx, long sy) {
    int fx = FN(lg_floor)(x), fy = FN(lg_floor)(y);
    int cx = fx + 1, cy = fy + 1;
    REAL t = x - (REAL)fx;
    REAL u = y - (REAL)fy;
    REAL omt = (REAL)1.f - t;
    REAL omu = (REAL)1.f - u;
    FN(lg_clamp_pair)(&fx, &cx, sx);
    FN(lg_clamp_pair)(&fy, &cy, sy);
    REAL v0 = img[(size_t)fx * sy + fy];
    REAL v1 = img[(size_t)cx * sy + fy];
    REAL v2 = img[(size_t)cx * sy + cy];
    REAL v3 = img[(size_t)fx * sy + cy];
    return LG_FMA(omt, LG_FMA(omu, v0, u * v3), t * LG_FMA(omu, v1, u * v2));
}

/* include/interp.h:60-123 triLerp */
static REAL FN(lg_trilerp)(const REAL *img, REAL x, REAL y, REAL z, long sx, long sy, long sz) {
    int fx = FN(lg_floor)(x), fy = FN(lg_floor)(y), fz = FN(lg_floor)(z);
    int cx = fx + 1, cy = fy + 1, cz = fz + 1;
    REAL t = x - (REAL)fx;
    REAL u = y - (REAL)fy;
    REAL v = z - (REAL)fz;
    REAL omt = (REAL)1.f - t;
    REAL omu = (REAL)1.f - u;
    REAL omv = (REAL)1.f - v;
    FN(lg_clamp_pair)(&fx, &cx, sx);
    FN(lg_clamp_pair)(&fy, &cy, sy);
    FN(lg_clamp_pair)(&fz, &cz, sz);
#define LG_AT(a, b, c) img[((size_t)(a) * sy + (b)) * sz + (c)]
    REAL v0 = LG_AT(fx, fy, fz), v1 = LG_AT(cx, fy, fz), v2 = LG_AT(cx, cy, fz), v3 = LG_AT(fx, cy, fz);
    REAL v4 = LG_AT(fx, fy, cz), v5 = LG_AT(cx, fy, cz), v6 = LG_AT(cx, cy, cz), v7 = LG_AT(fx, cy, cz);
#undef LG_AT
    return LG_FMA(omv, LG_FMA(omu, LG_FMA(omt, v0, t * v1), u * LG_FMA(omt, v3, t * v2)),
                  v * LG_FMA(omu, LG_FMA(omt, v4, t * v5), u * LG_FMA(omt, v7, t * v6)));
}

/* include/interp.h:128-204 biLerp_grad (clamp => always "inside") */
static void FN(lg_bilerp_grad)(REAL *gx, REAL *gy, const REAL *img, REAL x, REAL y, long sx, long sy) {
    int fx = FN(lg_floor)(x), fy = FN(lg_floor)(y);
    int cx = fx + 1, cy = fy + 1;
    REAL t = x - (REAL)fx;
    REAL u = y - (REAL)fy;
    FN(lg_clamp_pair)(&fx, &cx, sx);
    FN(lg_clamp_pair)(&fy, &cy, sy);
    REAL v0 = img[(size_t)fx * sy + fy];
    REAL v1 = img[(size_t)cx * sy + fy];
    REAL v2 = img[(size_t)cx * sy + cy];
    REAL v3 = img[(size_t)fx * sy + cy];
    *gx = LG_FMA(u, v2 - v3 - v1 + v0, v1 - v0);
    *gy = LG_FMA(t, v2 - v1 - v3 + v0, v3 - v0);
}

/* include/interp.h:206-327 triLerp_grad */
static void FN(lg_trilerp_grad)(REAL *gx, REAL *gy, REAL *gz, const REAL *img, REAL x, REAL y, REAL z,
                                long sx, long sy, long sz) {
    int fx = FN(lg_floor)(x), fy = FN(lg_floor)(y), fz = FN(lg_floor)(z);
    int cx = fx + 1, cy = fy + 1, cz = fz + 1;
    REAL t = x - (REAL)fx;
    REAL u = y - (REAL)fy;
    REAL v = z - (REAL)fz;
    REAL omt = (REAL)1.f - t;
    REAL omu = (REAL)1.f - u;
    REAL omv = (REAL)1.f - v;
    FN(lg_clamp_pair)(&fx, &cx, sx);
    FN(lg_clamp_pair)(&fy, &cy, sy);
    FN(lg_clamp_pair)(&fz, &cz, sz);
#define LG_AT(a, b, c) img[((size_t)(a) * sy + (b)) * sz + (c)]
    REAL v0 = LG_AT(fx, fy, fz), v1 = LG_AT(cx, fy, fz), v2 = LG_AT(cx, cy, fz), v3 = LG_AT(fx, cy, fz);
    REAL v4 = LG_AT(fx, fy, cz), v5 = LG_AT(cx, fy, cz), v6 = LG_AT(cx, cy, cz), v7 = LG_AT(fx, cy, cz);
#undef LG_AT
    *gx = LG_FMA(omv, LG_FMA(omu, v1 - v0, u * (v2 - v3)), v * LG_FMA(omu, v5 - v4, u * (v6 - v7)));
    *gy = LG_FMA(omv, LG_FMA(omt, v3 - v0, t * (v2 - v1)), v * LG_FMA(omt, v7 - v4, t * (v6 - v5)));
    *gz = LG_FMA(omu, LG_FMA(omt, v4 - v0, t * (v5 - v1)), u * LG_FMA(omt, v7 - v3, t * (v6 - v2)));
}

/* ---- splat ---------------------------------------------------------------- */

/* include/interp.h:403-425 atomicSplat 2D + :330-363 splat_neighbor.  The
 * weight flips (dy = 1-dy, dx = 1-dx) are sequential, exactly as written. */
static void FN(lg_splat2)(REAL *d, REAL mass, REAL x, REAL y, long w, long h) {
    int xi0 = FN(lg_floor)(x), yi0 = FN(lg_floor)(y);
    REAL dx = (REAL)1.f - (x - (REAL)xi0);
    REAL dy = (REAL)1.f - (y - (REAL)yi0);
    for (int xi = xi0; xi < xi0 + 2; xi++) {
        for (int yi = yi0; yi < yi0 + 2; yi++) {
            int i = FN(lg_clamp)(xi, w), j = FN(lg_clamp)(yi, h);
            REAL ww = dx * dy;
            d[(size_t)i * h + j] += ww * mass;
            dy = (REAL)1.f - dy;
        }
        dx = (REAL)1.f - dx;
    }
}

/* include/interp.h:426-454 atomicSplat 3D + :365-401 */
static void FN(lg_splat3)(REAL *d, REAL mass, REAL x, REAL y, REAL z, long w, long h, long l) {
    int xi0 = FN(lg_floor)(x), yi0 = FN(lg_floor)(y), zi0 = FN(lg_floor)(z);
    REAL dx = (REAL)1.f - (x - (REAL)xi0);
    REAL dy = (REAL)1.f - (y - (REAL)yi0);
    REAL dz = (REAL)1.f - (z - (REAL)zi0);
    for (int xi = xi0; xi < xi0 + 2; xi++) {
        for (int yi = yi0; yi < yi0 + 2; yi++) {
            for (int zi = zi0; zi < zi0 + 2; zi++) {
                int i = FN(lg_clamp)(xi, w), j = FN(lg_clamp)(yi, h), k = FN(lg_clamp)(zi, l);
                REAL ww = dx * dy * dz;
                d[((size_t)i * h + j) * l + k] += ww * mass;
                dz = (REAL)1.f - dz;
            }
            dy = (REAL)1.f - dy;
        }
        dx = (REAL)1.f - dx;
    }
}

/* The interpolation cores at caller-given points (test hook: pinned value by value against the
 * reference's own biLerp / triLerp / *_grad of include/interp.h compiled into oracle/_ref).
 * pts: (npts, dim); lerp: (npts); grad: (npts, dim). */
int FN(oracle_interp_points)(REAL *lerp, REAL *grad, const REAL *img, const REAL *pts, long npts, int dim, long sx,
                             long sy, long sz) {
    if (dim != 2 && dim != 3) return -1;
    for (long q = 0; q < npts; ++q) {
        if (dim == 2) {
            lerp[q] = FN(lg_bilerp)(img, pts[2 * q], pts[2 * q + 1], sx, sy);
            FN(lg_bilerp_grad)(&grad[2 * q], &grad[2 * q + 1], img, pts[2 * q], pts[2 * q + 1], sx, sy);
        } else {
            lerp[q] = FN(lg_trilerp)(img, pts[3 * q], pts[3 * q + 1], pts[3 * q + 2], sx, sy, sz);
            FN(lg_trilerp_grad)(&grad[3 * q], &grad[3 * q + 1], &grad[3 * q + 2], img, pts[3 * q], pts[3 * q + 1],
                                pts[3 * q + 2], sx, sy, sz);
        }
    }
    return 0;
}

/* The index rules at caller-given integer indices (test hooks: pinned against the reference's own include/extrap.h
 * compiled into oracle/_ref).  oracle_extrap_points: idx (npts, dim) of ANY integers; val = the clamped accessor
 * (get_value_safe<CLAMP>, extrap.h:110-192, through lg_clamp); grad = the clamped central differences of
 * include/diff.h:7-52 -- for a centre inside the grid by lg_grad_point itself, the function every jtv routine of this
 * oracle calls (rows a4-a7); for a centre outside (which no kernel produces) by the same expression on lg_clamp'ed
 * neighbours of the unclamped centre, as diff.h would evaluate it.  oracle_clamp_pairs: clampBackground
 * (extrap.h:46-57) on (floor, ceil) pairs. */
static inline void FN(lg_grad_point)(REAL *g, const REAL *a, int dim, long nx, long ny, long nz, long i, long j, long k);
int FN(oracle_extrap_points)(REAL *val, REAL *grad, const REAL *arr, const long *idx, long npts, int dim, long nx, long ny,
                             long nz) {
    if (dim != 2 && dim != 3) return -1;
    if (dim == 2) nz = 1;
    for (long q = 0; q < npts; ++q) {
        const long i = idx[dim * q], j = idx[dim * q + 1], k = dim == 3 ? idx[dim * q + 2] : 0;
        const long n[3] = {nx, ny, nz}, c[3] = {i, j, k};
        size_t at[3][3];   /* [axis][minus, centre, plus]: flat offset with that axis' index shifted, every index clamped */
        for (int a = 0; a < dim; ++a)
            for (int s = -1; s <= 1; ++s) {
                size_t off = 0;
                for (int b = 0; b < dim; ++b) {
                    const long r = FN(lg_clamp)((int)(c[b] + (b == a ? s : 0)), n[b]);
                    off = off * (size_t)n[b] + (size_t)r;
                }
                at[a][s + 1] = off;
            }
        val[q] = arr[at[0][1]];
        if (i >= 0 && i < nx && j >= 0 && j < ny && k >= 0 && k < nz)
            FN(lg_grad_point)(grad + dim * q, arr, dim, nx, ny, nz, i, j, k);
        else
            for (int a = 0; a < dim; ++a) grad[dim * q + a] = (REAL)0.5f * (arr[at[a][2]] - arr[at[a][0]]);
    }
    return 0;
}

int FN(oracle_clamp_pairs)(long *fl, long *ce, long size, long npts) {
    for (long q = 0; q < npts; ++q) {
        int f = (int)fl[q], c = (int)ce[q];
        FN(lg_clamp_pair)(&f, &c, size);
        fl[q] = f;
        ce[q] = c;
    }
    return 0;
}

/* ---- interp forward / backward -------------------------------------------- */

/* cuda/interp.cu:16-78 (kernels), :80-130 (host).  Position = fi + dt*u in
 * double (dt is double), narrowed to REAL at the lerp call. */
int FN(oracle_interp_forward)(REAL *out, const REAL *I, const REAL *u, double dt, int dim, long nn, long nc,
                              long nx, long ny, long nz, int broadcast_I) {
    if (dim != 2 && dim != 3) return -1;
    if (dim == 2) nz = 1;
    const size_t nvox = (size_t)nx * ny * nz;
    /* every output voxel is independent: (n, c, i) flattened into one loop so that OpenMP (bench.py's
     * cpu_baseline leg only) has nn*nc*nx work items */
    LG_PARALLEL_FOR
    for (long t = 0; t < nn * nc * nx; ++t) {
        const long n = t / (nc * nx), c = (t / nx) % nc, i = t % nx;
        const REAL *un = u + (size_t)n * dim * nvox;
        const REAL *Ic = (broadcast_I ? I : I + (size_t)n * nc * nvox) + (size_t)c * nvox;
        REAL *oc = out + (size_t)n * nc * nvox + (size_t)c * nvox;
        for (long j = 0; j < ny; ++j) {
            if (dim == 2) {
                size_t ix = (size_t)i * ny + j;
                double hx = LG_FMAD(dt, (double)un[ix], (double)(REAL)i);
                double hy = LG_FMAD(dt, (double)un[ix + nvox], (double)(REAL)j);
                oc[ix] = FN(lg_bilerp)(Ic, (REAL)hx, (REAL)hy, nx, ny);
            } else {
                for (long k = 0; k < nz; ++k) {
                    size_t ix = ((size_t)i * ny + j) * nz + k;
                    double hx = LG_FMAD(dt, (double)un[ix], (double)(REAL)i);
                    double hy = LG_FMAD(dt, (double)un[ix + nvox], (double)(REAL)j);
                    double hz = LG_FMAD(dt, (double)un[ix + 2 * nvox], (double)(REAL)k);
                    oc[ix] = FN(lg_trilerp)(Ic, (REAL)hx, (REAL)hy, (REAL)hz, nx, ny, nz);
                }
            }
        }
    }
    return 0;
}

/* One x-slab [i0, i1) of one channel of one batch item of interp_kernel_backward_{2,3}d (cuda/interp.cu:132-244):
 * the splat goes into `dIc` (a full channel plane), d_u is owned voxel by voxel. */
static void FN(lg_interp_backward_slab)(REAL *dIc, REAL *dun, const REAL *gc, const REAL *Ic, const REAL *un,
                                        double dt, int dim, long nx, long ny, long nz, long i0, long i1, int need_I,
                                        int need_u) {
    const size_t nvox = (size_t)nx * ny * nz;
    for (long i = i0; i < i1; ++i)
        for (long j = 0; j < ny; ++j)
            for (long k = 0; k < nz; ++k) {
                size_t ix = ((size_t)i * ny + j) * nz + k;
                REAL hx = (REAL)LG_FMAD(dt, (double)un[ix], (double)i);
                REAL hy = (REAL)LG_FMAD(dt, (double)un[ix + nvox], (double)j);
                REAL hz = 0;
                if (dim == 3) hz = (REAL)LG_FMAD(dt, (double)un[ix + 2 * nvox], (double)k);
                REAL diff = gc[ix];
                if (need_I) {
                    if (dim == 2)
                        FN(lg_splat2)(dIc, diff, hx, hy, nx, ny);
                    else
                        FN(lg_splat3)(dIc, diff, hx, hy, hz, nx, ny, nz);
                }
                if (need_u) {
                    REAL gx, gy, gz;
                    if (dim == 2) {
                        FN(lg_bilerp_grad)(&gx, &gy, Ic, hx, hy, nx, ny);
                        diff = (REAL)((double)diff * dt);
                        dun[ix] = LG_FMA(gx, diff, dun[ix]);
                        dun[ix + nvox] = LG_FMA(gy, diff, dun[ix + nvox]);
                    } else {
                        FN(lg_trilerp_grad)(&gx, &gy, &gz, Ic, hx, hy, hz, nx, ny, nz);
                        diff = (REAL)((double)diff * dt);
                        dun[ix] = LG_FMA(gx, diff, dun[ix]);
                        dun[ix + nvox] = LG_FMA(gy, diff, dun[ix + nvox]);
                        dun[ix + 2 * nvox] = LG_FMA(gz, diff, dun[ix + 2 * nvox]);
                    }
                }
            }
}

/* cuda/interp.cu:132-244 (kernels), :246-313 (host: both outputs always
 * allocated as zeros and returned). */
int FN(oracle_interp_backward)(REAL *d_I, REAL *d_u, const REAL *go, const REAL *I, const REAL *u, double dt,
                               int dim, long nn, long nc, long nx, long ny, long nz, int broadcast_I,
                               int need_I, int need_u) {
    if (dim != 2 && dim != 3) return -1;
    if (dim == 2) nz = 1;
    const size_t nvox = (size_t)nx * ny * nz;
    const size_t nI = (broadcast_I ? 1 : (size_t)nn) * nc * nvox;
    memset(d_I, 0, nI * sizeof(REAL));
    memset(d_u, 0, (size_t)nn * dim * nvox * sizeof(REAL));
    /* cpu_baseline leg only (bench.py raises the thread count; the tests run with one thread): with more threads than
     * batch items every (item, channel) is cut into x-slabs that splat into private planes, which are then added
     * onto d_I in slab order -- a deterministic summation order, but not the sequential one (the reference's atomic
     * order is unspecified as well).  d_u is owned voxel by voxel and does not depend on the split. */
    const long slabs = (lg_oracle_threads > nn && nn > 0 && need_I) ? (lg_oracle_threads + nn - 1) / nn : 1;
    if (slabs > 1 && nx >= 2 * slabs) {
        const long ntask = nn * slabs;
        REAL *priv = (REAL *)calloc((size_t)ntask * nvox, sizeof(REAL));
        if (!priv) return -2;
        for (long c = 0; c < nc; ++c) {
            if (c) memset(priv, 0, (size_t)ntask * nvox * sizeof(REAL));
            LG_PARALLEL_FOR_IF(1)
            for (long t = 0; t < ntask; ++t) {
                const long n = t / slabs, sl = t % slabs;
                const long i0 = nx * sl / slabs, i1 = nx * (sl + 1) / slabs;
                const REAL *In = broadcast_I ? I : I + (size_t)n * nc * nvox;
                FN(lg_interp_backward_slab)(priv + (size_t)t * nvox, d_u + (size_t)n * dim * nvox,
                                            go + ((size_t)n * nc + c) * nvox, In + (size_t)c * nvox,
                                            u + (size_t)n * dim * nvox, dt, dim, nx, ny, nz, i0, i1, need_I, need_u);
            }
            LG_PARALLEL_FOR
            for (long x = 0; x < (long)nvox; ++x)
                for (long t = 0; t < ntask; ++t) {
                    REAL *dIc = (broadcast_I ? d_I : d_I + (size_t)(t / slabs) * nc * nvox) + (size_t)c * nvox;
                    dIc[x] += priv[(size_t)t * nvox + x];
                }
        }
        free(priv);
        return 0;
    }
    /* a batch item owns its d_u, and its d_I unless I is broadcast: OpenMP over n (cpu_baseline leg only)
     * leaves every sum in the sequential order */
    LG_PARALLEL_FOR_IF(!broadcast_I || !need_I)
    for (long n = 0; n < nn; ++n) {
        const REAL *un = u + (size_t)n * dim * nvox;
        REAL *dun = d_u + (size_t)n * dim * nvox;
        const REAL *In = broadcast_I ? I : I + (size_t)n * nc * nvox;
        REAL *dIn = broadcast_I ? d_I : d_I + (size_t)n * nc * nvox;
        const REAL *gon = go + (size_t)n * nc * nvox;
        for (long c = 0; c < nc; ++c)
            FN(lg_interp_backward_slab)(dIn + (size_t)c * nvox, dun, gon + (size_t)c * nvox, In + (size_t)c * nvox, un,
                                        dt, dim, nx, ny, nz, 0, nx, need_I, need_u);
    }
    return 0;
}

/* cuda/interp.cu:317-381 + include/interp.h:459-544.  2D only.  The kernel
 * passes `out` (not the per-channel pointer) to the point routine, so every
 * (n, c) accumulates into plane 0 of the output; restated as coded. */
int FN(oracle_interp_hessian_diagonal_image)(REAL *out, const REAL *u, double dt, long nI, long nn, long nc,
                                             long nx, long ny) {
    const size_t nxy = (size_t)nx * ny;
    memset(out, 0, (size_t)nI * nc * nxy * sizeof(REAL));
    for (long n = 0; n < nn; ++n) {
        const REAL *un = u + (size_t)n * 2 * nxy;
        for (long i = 0; i < nx; ++i)
            for (long j = 0; j < ny; ++j) {
                size_t ix = (size_t)i * ny + j;
                REAL x = (REAL)LG_FMAD(dt, (double)un[ix], (double)(REAL)i);
                REAL y = (REAL)LG_FMAD(dt, (double)un[ix + nxy], (double)(REAL)j);
                int fx = FN(lg_floor)(x), fy = FN(lg_floor)(y);
                int cx = fx + 1, cy = fy + 1;
                REAL t = x - (REAL)fx, uu = y - (REAL)fy;
                REAL omt = (REAL)1.f - t, omu = (REAL)1.f - uu;
                FN(lg_clamp_pair)(&fx, &cx, nx);
                FN(lg_clamp_pair)(&fy, &cy, ny);
                REAL w0 = omt * omu, w1 = t * omu, w2 = t * uu, w3 = omt * uu;
                for (long c = 0; c < nc; ++c) {
                    out[(size_t)fx * ny + fy] += w0 * w0;
                    out[(size_t)cx * ny + fy] += w1 * w1;
                    out[(size_t)cx * ny + cy] += w2 * w2;
                    out[(size_t)fx * ny + cy] += w3 * w3;
                }
            }
    }
    return 0;
}

/* ---- finite differences --------------------------------------------------- */

/* include/diff.h:7-76 grad_point with get_value_safe<CLAMP> (extrap.h:110-192):
 * 0.5*(f(clamp(i+1)) - f(clamp(i-1))) per axis.  g[0..dim-1]. */
static inline void FN(lg_grad_point)(REAL *g, const REAL *a, int dim, long nx, long ny, long nz, long i, long j,
                                     long k) {
    if (dim == 2) {
        long ip = i + 1 < nx ? i + 1 : nx - 1, im = i - 1 < 0 ? 0 : i - 1;
        long jp = j + 1 < ny ? j + 1 : ny - 1, jm = j - 1 < 0 ? 0 : j - 1;
        g[0] = (REAL)0.5f * (a[(size_t)ip * ny + j] - a[(size_t)im * ny + j]);
        g[1] = (REAL)0.5f * (a[(size_t)i * ny + jp] - a[(size_t)i * ny + jm]);
    } else {
        long ip = i + 1 < nx ? i + 1 : nx - 1, im = i - 1 < 0 ? 0 : i - 1;
        long jp = j + 1 < ny ? j + 1 : ny - 1, jm = j - 1 < 0 ? 0 : j - 1;
        long kp = k + 1 < nz ? k + 1 : nz - 1, km = k - 1 < 0 ? 0 : k - 1;
        g[0] = (REAL)0.5f * (a[((size_t)ip * ny + j) * nz + k] - a[((size_t)im * ny + j) * nz + k]);
        g[1] = (REAL)0.5f * (a[((size_t)i * ny + jp) * nz + k] - a[((size_t)i * ny + jm) * nz + k]);
        g[2] = (REAL)0.5f * (a[((size_t)i * ny + j) * nz + kp] - a[((size_t)i * ny + j) * nz + km]);
    }
}

/* cuda/diff.cu:17-127 (kernels), :129-185 (host).  `v` is the differentiated
 * field (first Python argument), `w` the contracted one. */
int FN(oracle_jtv_forward)(REAL *out, const REAL *v, const REAL *w, int displacement, int transpose, int dim,
                           long nn, long nc, long nx, long ny, long nz) {
    if (dim != 2 && dim != 3) return -1;
    if (dim == 2) nz = 1;
    if ((displacement || transpose) && nc != dim) return -1;
    const size_t nvox = (size_t)nx * ny * nz;
    LG_PARALLEL_FOR
    for (long t = 0; t < nn * nx; ++t) {
        const long n = t / nx, i = t % nx;
        const REAL *vn = v + (size_t)n * nc * nvox;
        const REAL *wn = w + (size_t)n * dim * nvox;
        REAL *on = out + (size_t)n * nc * nvox;
        {
            for (long j = 0; j < ny; ++j)
                for (long k = 0; k < nz; ++k) {
                    REAL g[3];
                    size_t ix = ((size_t)i * ny + j) * nz + k;
                    if (transpose) {
                        /* out[d] = sum_c (d_d v_c + delta) w_c, accumulated c = 0,1,2 */
                        for (int c = 0; c < dim; ++c) {
                            FN(lg_grad_point)(g, vn + (size_t)c * nvox, dim, nx, ny, nz, i, j, k);
                            if (displacement) g[c] = g[c] + (REAL)1.0;
                            REAL wc = wn[ix + (size_t)c * nvox];
                            for (int d = 0; d < dim; ++d) {
                                if (c == 0)
                                    on[ix + (size_t)d * nvox] = g[d] * wc;
                                else
                                    on[ix + (size_t)d * nvox] = LG_FMA(g[d], wc, on[ix + (size_t)d * nvox]);
                            }
                        }
                    } else {
                        for (long c = 0; c < nc; ++c) {
                            FN(lg_grad_point)(g, vn + (size_t)c * nvox, dim, nx, ny, nz, i, j, k);
                            if (displacement && c < dim) g[c] = g[c] + (REAL)1.0;
                            REAL s = LG_FMA(g[0], wn[ix], g[1] * wn[ix + nvox]);
                            if (dim == 3) s = LG_FMA(g[2], wn[ix + 2 * nvox], s);
                            on[ix + (size_t)c * nvox] = s;
                        }
                    }
                }
        }
    }
    return 0;
}

/* Adjoint of the clamped central difference along one axis applied to the
 * product field p[m] = a[m]*b[m]: the reference's three-case border formula
 * (cuda/diff.cu:224-248, 334-391, 560-573, 603-620).  `s` is the element
 * stride of the axis, `pos`/`len` the coordinate and extent on it; ia/ib are
 * the linear indices of the centre voxel inside a and b. */
static inline REAL FN(lg_dT_term)(const REAL *a, size_t ia, const REAL *b, size_t ib, size_t s, long pos,
                                  long len) {
    if (len == 1) return (REAL)0; /* see below */
    if (pos == 0) return (REAL)(-.5) * LG_FMA(a[ia], b[ib], a[ia + s] * b[ib + s]);
    if (pos == len - 1) return (REAL)(.5) * LG_FMA(a[ia], b[ib], a[ia - s] * b[ib - s]);
    return (REAL)(-.5) * LG_FMA(a[ia + s], b[ib + s], -(a[ia - s] * b[ib - s]));
}
/* An axis of extent 1: the clamped difference along it is identically zero and
 * so is its adjoint.  The reference's `pos == 0` case would read a[ia + s] there,
 * which is the next channel or past the end of the allocation (undefined); the
 * oracle and the HIP kernels both define it as 0 (deviations).
 * cuda/diff.cu:345-356 and :597-600 write the i==0 / j==0 / k==0 case of the 3D
 * kernels with the +stride product first; here the centre product is the fused
 * one in every case (which product nvcc fuses is its choice). */

/* cuda/diff.cu:187-473 (kernels), :475-540 (host; need_v/need_w forced true). */
int FN(oracle_jtv_backward)(REAL *d_v, REAL *d_w, const REAL *go, const REAL *v, const REAL *w, int displacement,
                            int transpose, int dim, long nn, long nc, long nx, long ny, long nz) {
    if (dim != 2 && dim != 3) return -1;
    if (dim == 2) nz = 1;
    if ((displacement || transpose) && nc != dim) return -1;
    const size_t nvox = (size_t)nx * ny * nz;
    const size_t str[3] = {(size_t)ny * nz, (size_t)nz, 1};
    const long len[3] = {nx, ny, nz};
    memset(d_v, 0, (size_t)nn * nc * nvox * sizeof(REAL));
    memset(d_w, 0, (size_t)nn * dim * nvox * sizeof(REAL));
    REAL g[3];
    for (long n = 0; n < nn; ++n) {
        const REAL *vn = v + (size_t)n * nc * nvox;
        const REAL *wn = w + (size_t)n * dim * nvox;
        const REAL *gon = go + (size_t)n * nc * nvox;
        REAL *dvn = d_v + (size_t)n * nc * nvox;
        REAL *dwn = d_w + (size_t)n * dim * nvox;
        for (long i = 0; i < nx; ++i)
            for (long j = 0; j < ny; ++j)
                for (long k = 0; k < nz; ++k) {
                    const long pos[3] = {i, j, k};
                    size_t ix = ((size_t)i * ny + j) * nz + k;
                    if (transpose) {
                        /* d_w[c] = sum_d (d_d v_c + delta) go_d ; d_v[c] += sum_d D_d^T (w_c go_d) */
                        for (int c = 0; c < dim; ++c) {
                            FN(lg_grad_point)(g, vn + (size_t)c * nvox, dim, nx, ny, nz, i, j, k);
                            if (displacement) g[c] = g[c] + (REAL)1.0;
                            REAL s = LG_FMA(g[0], gon[ix], g[1] * gon[ix + nvox]);
                            if (dim == 3) s = LG_FMA(g[2], gon[ix + 2 * nvox], s);
                            dwn[ix + (size_t)c * nvox] += s;
                        }
                        for (int d = 0; d < dim; ++d)     /* axis, in reference order x,y,z */
                            for (int c = 0; c < dim; ++c) /* component of v */
                                dvn[ix + (size_t)c * nvox] +=
                                    FN(lg_dT_term)(wn + (size_t)c * nvox, ix, gon + (size_t)d * nvox, ix,
                                                   str[d], pos[d], len[d]);
                    } else {
                        for (long c = 0; c < nc; ++c) {
                            FN(lg_grad_point)(g, vn + (size_t)c * nvox, dim, nx, ny, nz, i, j, k);
                            if (displacement && c < dim) g[c] = g[c] + (REAL)1.0;
                            REAL goc = gon[ix + (size_t)c * nvox];
                            for (int d = 0; d < dim; ++d) dwn[ix + (size_t)d * nvox] = LG_FMA(g[d], goc, dwn[ix + (size_t)d * nvox]);
                            for (int d = 0; d < dim; ++d)
                                dvn[ix + (size_t)c * nvox] +=
                                    FN(lg_dT_term)(wn + (size_t)d * nvox, ix, gon + (size_t)c * nvox, ix,
                                                   str[d], pos[d], len[d]);
                        }
                    }
                }
    }
    return 0;
}

/* cuda/diff.cu:546-632 (kernels), :634-672 (host): out[c] = sum_d D_d^T (w_d z_c) */
int FN(oracle_jtv_adjoint_forward)(REAL *out, const REAL *z, const REAL *w, int dim, long nn, long nc, long nx,
                                   long ny, long nz) {
    if (dim != 2 && dim != 3) return -1;
    if (dim == 2) nz = 1;
    const size_t nvox = (size_t)nx * ny * nz;
    const size_t str[3] = {(size_t)ny * nz, (size_t)nz, 1};
    const long len[3] = {nx, ny, nz};
    memset(out, 0, (size_t)nn * nc * nvox * sizeof(REAL));
    for (long n = 0; n < nn; ++n) {
        const REAL *zn = z + (size_t)n * nc * nvox;
        const REAL *wn = w + (size_t)n * dim * nvox;
        REAL *on = out + (size_t)n * nc * nvox;
        for (long c = 0; c < nc; ++c)
            for (long i = 0; i < nx; ++i)
                for (long j = 0; j < ny; ++j)
                    for (long k = 0; k < nz; ++k) {
                        const long pos[3] = {i, j, k};
                        size_t ix = ((size_t)i * ny + j) * nz + k;
                        for (int d = 0; d < dim; ++d)
                            on[ix + (size_t)c * nvox] +=
                                FN(lg_dT_term)(wn + (size_t)d * nvox, ix, zn + (size_t)c * nvox, ix,
                                               str[d], pos[d], len[d]);
                    }
    }
    return 0;
}

/* cuda/diff.cu:674-780 (kernels), :783-835 (host; nc == dim hard-coded,
 * need_* forced true).  v is the `z` argument of the forward. */
int FN(oracle_jtv_adjoint_backward)(REAL *d_v, REAL *d_w, const REAL *go, const REAL *v, const REAL *w, int dim,
                                    long nn, long nx, long ny, long nz) {
    if (dim != 2 && dim != 3) return -1;
    if (dim == 2) nz = 1;
    const size_t nvox = (size_t)nx * ny * nz;
    memset(d_v, 0, (size_t)nn * dim * nvox * sizeof(REAL));
    memset(d_w, 0, (size_t)nn * dim * nvox * sizeof(REAL));
    REAL g[3];
    for (long n = 0; n < nn; ++n) {
        const REAL *vn = v + (size_t)n * dim * nvox;
        const REAL *wn = w + (size_t)n * dim * nvox;
        const REAL *gon = go + (size_t)n * dim * nvox;
        REAL *dvn = d_v + (size_t)n * dim * nvox;
        REAL *dwn = d_w + (size_t)n * dim * nvox;
        for (long i = 0; i < nx; ++i)
            for (long j = 0; j < ny; ++j)
                for (long k = 0; k < nz; ++k) {
                    size_t ix = ((size_t)i * ny + j) * nz + k;
                    for (int c = 0; c < dim; ++c) {
                        FN(lg_grad_point)(g, gon + (size_t)c * nvox, dim, nx, ny, nz, i, j, k);
                        REAL vc = vn[ix + (size_t)c * nvox];
                        for (int d = 0; d < dim; ++d) {
                            if (c == 0)
                                dwn[ix + (size_t)d * nvox] = g[d] * vc;
                            else
                                dwn[ix + (size_t)d * nvox] = LG_FMA(g[d], vc, dwn[ix + (size_t)d * nvox]);
                        }
                        REAL s = LG_FMA(g[0], wn[ix], g[1] * wn[ix + nvox]);
                        if (dim == 3) s = LG_FMA(g[2], wn[ix + 2 * nvox], s);
                        dvn[ix + (size_t)c * nvox] += s;
                    }
                }
    }
    return 0;
}

/* ---- fluid metric operator ------------------------------------------------ */

/* cuda/metric.cu:14-18 */
static inline REAL FN(lg_safe_sqrt)(REAL x) {
    if ((double)x < 1e-8) return (REAL)1e-4;
    return (REAL)LG_SQRT(x);
}

/* cuda/metric.cu:162-306 (kernels), :308-355 (host).  Fm is interleaved
 * complex of shape (nn, dim, nx, ny[, nzc], 2); modified in place.
 * alpha/beta/gamma are double: lambda, l_cc, l_cd are evaluated in double and
 * narrowed to REAL exactly where the reference assigns to a Real. */
int FN(oracle_fluid_operator)(REAL *Fm, int inverse, const REAL *cosX, const REAL *sinX, const REAL *cosY,
                              const REAL *sinY, const REAL *cosZ, const REAL *sinZ, double alpha, double beta,
                              double gamma, int dim, long nn, long nx, long ny, long nz) {
    if (dim != 2 && dim != 3) return -1;
    if (dim == 2) {
        const size_t nxy = 2 * (size_t)nx * ny;
        for (long i = 0; i < nx; ++i)
            for (long j = 0; j < ny; ++j) {
                const REAL wx = cosX[i], wy = cosY[j];
                const REAL lambda = (REAL)LG_FMAD(alpha, (double)(wx + wy), gamma);
                REAL l00 = (REAL)LG_FMAD(-beta, (double)wx, (double)lambda);
                REAL l11 = (REAL)LG_FMAD(-beta, (double)wy, (double)lambda);
                REAL l10 = (REAL)(beta * (double)sinX[i] * (double)sinY[j]);
                REAL L00 = LG_FMA(l00, l00, l10 * l10);
                REAL L10 = LG_FMA(l00, l10, l10 * l11);
                REAL L11 = LG_FMA(l11, l11, l10 * l10);
                REAL ooG00 = 0, G10 = 0, ooG11 = 0;
                if (inverse) { /* metric.cu:20-45 */
                    ooG00 = (REAL)(1. / (double)FN(lg_safe_sqrt)(L00));
                    G10 = L10 * ooG00;
                    ooG11 = LG_FMA(-G10, G10, L11);
                    ooG11 = (REAL)(1. / (double)FN(lg_safe_sqrt)(ooG11));
                }
                size_t ix = 2 * ((size_t)j + (size_t)i * ny);
                for (long n = 0; n < nn; ++n, ix += 2 * nxy) {
                    size_t iy = ix + nxy;
                    for (int ri = 0; ri < 2; ++ri) {
                        REAL bX = Fm[ix + ri], bY = Fm[iy + ri];
                        if (inverse) { /* metric.cu:80-101 */
                            REAL y0 = bX * ooG00;
                            REAL y1 = LG_FMA(-G10, y0, bY) * ooG11;
                            bY = y1 * ooG11;
                            bX = LG_FMA(-G10, bY, y0) * ooG00;
                        } else { /* metric.cu:132-143 */
                            REAL x = LG_FMA(L00, bX, L10 * bY);
                            bY = LG_FMA(L10, bX, L11 * bY);
                            bX = x;
                        }
                        Fm[ix + ri] = bX;
                        Fm[iy + ri] = bY;
                    }
                }
            }
        return 0;
    }
    const size_t nxyz = 2 * (size_t)nx * ny * nz;
    LG_PARALLEL_FOR
    for (long i = 0; i < nx; ++i)
        for (long j = 0; j < ny; ++j)
            for (long k = 0; k < nz; ++k) {
                const REAL wx = cosX[i], wy = cosY[j], wz = cosZ[k];
                const REAL lambda = (REAL)LG_FMAD(alpha, (double)(wx + wy + wz), gamma);
                REAL l00 = (REAL)LG_FMAD(-beta, (double)wx, (double)lambda);
                REAL l11 = (REAL)LG_FMAD(-beta, (double)wy, (double)lambda);
                REAL l22 = (REAL)LG_FMAD(-beta, (double)wz, (double)lambda);
                REAL l10 = (REAL)(beta * (double)sinX[i] * (double)sinY[j]);
                REAL l20 = (REAL)(beta * (double)sinX[i] * (double)sinZ[k]);
                REAL l21 = (REAL)(beta * (double)sinY[j] * (double)sinZ[k]);
                REAL L00 = LG_FMA(l20, l20, LG_FMA(l00, l00, l10 * l10));
                REAL L10 = LG_FMA(l20, l21, LG_FMA(l00, l10, l10 * l11));
                REAL L11 = LG_FMA(l21, l21, LG_FMA(l10, l10, l11 * l11));
                REAL L20 = LG_FMA(l20, l22, LG_FMA(l00, l20, l10 * l21));
                REAL L21 = LG_FMA(l21, l22, LG_FMA(l10, l20, l11 * l21));
                REAL L22 = LG_FMA(l22, l22, LG_FMA(l20, l20, l21 * l21));
                REAL ooG00 = 0, G10 = 0, ooG11 = 0, G20 = 0, G21 = 0, ooG22 = 0;
                if (inverse) { /* metric.cu:47-78 */
                    ooG00 = (REAL)(1. / (double)FN(lg_safe_sqrt)(L00));
                    G10 = L10 * ooG00;
                    G20 = L20 * ooG00;
                    ooG11 = LG_FMA(-G10, G10, L11);
                    ooG11 = (REAL)(1. / (double)FN(lg_safe_sqrt)(ooG11));
                    G21 = LG_FMA(-G20, G10, L21) * ooG11;
                    ooG22 = LG_FMA(-G21, G21, LG_FMA(-G20, G20, L22));
                    ooG22 = (REAL)(1. / (double)FN(lg_safe_sqrt)(ooG22));
                }
                size_t ix = 2 * ((size_t)j + (size_t)i * ny) * nz + 2 * (size_t)k;
                for (long n = 0; n < nn; ++n, ix += 3 * nxyz) {
                    size_t iy = ix + nxyz, iz = iy + nxyz;
                    for (int ri = 0; ri < 2; ++ri) {
                        REAL bX = Fm[ix + ri], bY = Fm[iy + ri], bZ = Fm[iz + ri];
                        if (inverse) { /* metric.cu:103-130 */
                            REAL y0 = bX * ooG00;
                            REAL y1 = LG_FMA(-G10, y0, bY) * ooG11;
                            REAL y2 = LG_FMA(-G21, y1, LG_FMA(-G20, y0, bZ)) * ooG22;
                            bZ = y2 * ooG22;
                            bY = LG_FMA(-G21, bZ, y1) * ooG11;
                            bX = LG_FMA(-G20, bZ, LG_FMA(-G10, bY, y0)) * ooG00;
                        } else { /* metric.cu:145-160 */
                            REAL x = LG_FMA(L20, bZ, LG_FMA(L00, bX, L10 * bY));
                            REAL y = LG_FMA(L21, bZ, LG_FMA(L10, bX, L11 * bY));
                            bZ = LG_FMA(L22, bZ, LG_FMA(L20, bX, L21 * bY));
                            bX = x;
                            bY = y;
                        }
                        Fm[ix + ri] = bX;
                        Fm[iy + ri] = bY;
                        Fm[iz + ri] = bZ;
                    }
                }
            }
    return 0;
}

/* ---- affine interpolation ------------------------------------------------- */

/* cuda/affine.cu:23-112 (kernels), :114-169 (host): h = A (x - o) + T + o,
 * o = (n-1)/2, evaluated per voxel (no incremental update). */
int FN(oracle_affine_interp_forward)(REAL *out, const REAL *I, const REAL *A, const REAL *T, int dim, long nn,
                                     long nc, long nx, long ny, long nz, int broadcast_I) {
    if (dim != 2 && dim != 3) return -1;
    if (dim == 2) nz = 1;
    const size_t nvox = (size_t)nx * ny * nz;
    const REAL ox = (REAL)(.5 * (double)(REAL)(nx - 1));
    const REAL oy = (REAL)(.5 * (double)(REAL)(ny - 1));
    const REAL oz = (REAL)(.5 * (double)(REAL)(nz - 1));
    for (long n = 0; n < nn; ++n) {
        const REAL *An = A + (size_t)n * dim * dim;
        const REAL *Tn = T + (size_t)n * dim;
        const REAL *In = broadcast_I ? I : I + (size_t)n * nc * nvox;
        REAL *on = out + (size_t)n * nc * nvox;
        for (long c = 0; c < nc; ++c) {
            const REAL *Ic = In + (size_t)c * nvox;
            REAL *oc = on + (size_t)c * nvox;
            for (long i = 0; i < nx; ++i) {
                REAL fi = (REAL)i - ox;
                for (long j = 0; j < ny; ++j) {
                    REAL fj = (REAL)j - oy;
                    if (dim == 2) {
                        REAL hx = LG_FMA(An[0], fi, An[1] * fj) + Tn[0] + ox;
                        REAL hy = LG_FMA(An[2], fi, An[3] * fj) + Tn[1] + oy;
                        oc[(size_t)i * ny + j] = FN(lg_bilerp)(Ic, hx, hy, nx, ny);
                    } else {
                        for (long k = 0; k < nz; ++k) {
                            REAL fk = (REAL)k - oz;
                            REAL hx = LG_FMA(An[2], fk, LG_FMA(An[0], fi, An[1] * fj)) + Tn[0] + ox;
                            REAL hy = LG_FMA(An[5], fk, LG_FMA(An[3], fi, An[4] * fj)) + Tn[1] + oy;
                            REAL hz = LG_FMA(An[8], fk, LG_FMA(An[6], fi, An[7] * fj)) + Tn[2] + oz;
                            oc[((size_t)i * ny + j) * nz + k] = FN(lg_trilerp)(Ic, hx, hy, hz, nx, ny, nz);
                        }
                    }
                }
            }
        }
    }
    return 0;
}

/* cpu/affine.cpp:11-127 (loops), :129-169 (host): the reference's own CPU
 * path, which advances h incrementally along the fastest axis.  Used to pin
 * the lerp/floor/clamp core against oracle/_ref (the reference file compiled
 * as it lies). */
int FN(oracle_affine_interp_forward_cpuref)(REAL *out, const REAL *I, const REAL *A, const REAL *T, int dim,
                                            long nn, long nc, long nx, long ny, long nz, int broadcast_I) {
    if (dim != 2 && dim != 3) return -1;
    if (dim == 2) nz = 1;
    const size_t nvox = (size_t)nx * ny * nz;
    const REAL ox = (REAL)(.5 * (double)(REAL)(nx - 1));
    const REAL oy = (REAL)(.5 * (double)(REAL)(ny - 1));
    const REAL oz = (REAL)(.5 * (double)(REAL)(nz - 1));
    const REAL *In = I;
    REAL *outn = out;
    for (long n = 0; n < nn; ++n) {
        const REAL *An = A + (size_t)n * dim * dim;
        const REAL *Tn = T + (size_t)n * dim;
        if (broadcast_I) In = I;
        for (long c = 0; c < nc; ++c) {
            for (long i = 0; i < nx; ++i) {
                REAL fi = (REAL)i - ox;
                if (dim == 2) {
                    size_t ix = (size_t)i * ny;
                    REAL hx = LG_FMA(An[0], fi, -(An[1] * oy)) + Tn[0] + ox;
                    REAL hy = LG_FMA(An[2], fi, -(An[3] * oy)) + Tn[1] + oy;
                    for (long j = 0; j < ny; ++j, ++ix, hx += An[1], hy += An[3])
                        outn[ix] = FN(lg_bilerp)(In, hx, hy, nx, ny);
                } else {
                    for (long j = 0; j < ny; ++j) {
                        size_t ix = ((size_t)i * ny + j) * nz;
                        REAL fj = (REAL)j - oy;
                        REAL hx = LG_FMA(-An[2], oz, LG_FMA(An[0], fi, An[1] * fj)) + Tn[0] + ox;
                        REAL hy = LG_FMA(-An[5], oz, LG_FMA(An[3], fi, An[4] * fj)) + Tn[1] + oy;
                        REAL hz = LG_FMA(-An[8], oz, LG_FMA(An[6], fi, An[7] * fj)) + Tn[2] + oz;
                        for (long k = 0; k < nz; ++k, ++ix, hx += An[2], hy += An[5], hz += An[8])
                            outn[ix] = FN(lg_trilerp)(In, hx, hy, hz, nx, ny, nz);
                    }
                }
            }
            outn += nvox;
            In += nvox;
        }
    }
    return 0;
}

/* cuda/affine.cu:171-536 (kernels), :538-610 (host).  One 16x32 block per
 * (n, c): thread (ii, jj) accumulates voxels i = ii (mod 16), j = jj (mod 32),
 * all k, then the 512 partials are tree-reduced (256, 128, ... 1).  That order
 * is deterministic in the reference and is restated here; the cross-channel
 * atomicAdd (nc > 1) is taken in ascending c.  d_A / d_T may be NULL when not
 * needed (the reference returns size-0 tensors). */
int FN(oracle_affine_interp_backward)(REAL *d_I, REAL *d_A, REAL *d_T, const REAL *go, const REAL *I,
                                      const REAL *A, const REAL *T, int dim, long nn, long nc, long nx, long ny,
                                      long nz, int broadcast_I, int need_I, int need_A, int need_T) {
    if (dim != 2 && dim != 3) return -1;
    if (dim == 2) nz = 1;
    const size_t nvox = (size_t)nx * ny * nz;
    const int TX = 16, TY = 32, NT = 512;
    if (need_I) memset(d_I, 0, (broadcast_I ? 1 : (size_t)nn) * nc * nvox * sizeof(REAL));
    if (need_A) memset(d_A, 0, (size_t)nn * dim * dim * sizeof(REAL));
    if (need_T) memset(d_T, 0, (size_t)nn * dim * sizeof(REAL));
    const REAL ox = (REAL)(.5 * (double)(REAL)(nx - 1));
    const REAL oy = (REAL)(.5 * (double)(REAL)(ny - 1));
    const REAL oz = (REAL)(.5 * (double)(REAL)(nz - 1));
    REAL(*part)[12] = (REAL(*)[12])malloc(sizeof(REAL[12]) * NT);
    if (!part) return -2;
    for (long n = 0; n < nn; ++n) {
        const REAL *An = A + (size_t)n * dim * dim;
        const REAL *Tn = T + (size_t)n * dim;
        for (long c = 0; c < nc; ++c) {
            const REAL *gon = go + ((size_t)n * nc + c) * nvox;
            const REAL *In = I + ((broadcast_I ? 0 : (size_t)n * nc) + c) * nvox;
            REAL *dIn = need_I ? d_I + ((broadcast_I ? 0 : (size_t)n * nc) + c) * nvox : NULL;
            for (int ii = 0; ii < TX; ++ii)
                for (int jj = 0; jj < TY; ++jj) {
                    REAL *p = part[ii * TY + jj];
                    for (int q = 0; q < 12; ++q) p[q] = 0;
                    for (long i = ii; i < nx; i += TX) {
                        REAL fi = (REAL)i - ox;
                        for (long j = jj; j < ny; j += TY) {
                            REAL fj = (REAL)j - oy;
                            for (long k = 0; k < nz; ++k) {
                                size_t ix = ((size_t)i * ny + j) * nz + k;
                                REAL diff = gon[ix];
                                REAL gx, gy, gz = 0, fk = 0;
                                if (dim == 2) {
                                    REAL hx = LG_FMA(An[0], fi, An[1] * fj) + Tn[0] + ox;
                                    REAL hy = LG_FMA(An[2], fi, An[3] * fj) + Tn[1] + oy;
                                    if (need_I) FN(lg_splat2)(dIn, diff, hx, hy, nx, ny);
                                    if (!(need_A || need_T)) continue;
                                    FN(lg_bilerp_grad)(&gx, &gy, In, hx, hy, nx, ny);
                                } else {
                                    fk = (REAL)k - oz;
                                    REAL hx = LG_FMA(An[2], fk, LG_FMA(An[0], fi, An[1] * fj)) + Tn[0] + ox;
                                    REAL hy = LG_FMA(An[5], fk, LG_FMA(An[3], fi, An[4] * fj)) + Tn[1] + oy;
                                    REAL hz = LG_FMA(An[8], fk, LG_FMA(An[6], fi, An[7] * fj)) + Tn[2] + oz;
                                    if (need_I) FN(lg_splat3)(dIn, diff, hx, hy, hz, nx, ny, nz);
                                    if (!(need_A || need_T)) continue;
                                    FN(lg_trilerp_grad)(&gx, &gy, &gz, In, hx, hy, hz, nx, ny, nz);
                                }
                                gx *= diff;
                                gy *= diff;
                                gz *= diff;
                                if (dim == 2) {
                                    if (need_A) {
                                        p[0] = LG_FMA(gx, fi, p[0]); p[1] = LG_FMA(gx, fj, p[1]);
                                        p[2] = LG_FMA(gy, fi, p[2]); p[3] = LG_FMA(gy, fj, p[3]);
                                    }
                                    if (need_T) { p[9] += gx; p[10] += gy; }
                                } else {
                                    if (need_A) {
                                        p[0] = LG_FMA(gx, fi, p[0]); p[1] = LG_FMA(gx, fj, p[1]); p[2] = LG_FMA(gx, fk, p[2]);
                                        p[3] = LG_FMA(gy, fi, p[3]); p[4] = LG_FMA(gy, fj, p[4]); p[5] = LG_FMA(gy, fk, p[5]);
                                        p[6] = LG_FMA(gz, fi, p[6]); p[7] = LG_FMA(gz, fj, p[7]); p[8] = LG_FMA(gz, fk, p[8]);
                                    }
                                    if (need_T) { p[9] += gx; p[10] += gy; p[11] += gz; }
                                }
                            }
                        }
                    }
                }
            if (need_A || need_T) {
                for (int N = NT / 2; N >= 1; N /= 2)
                    for (int tid = 0; tid < N; ++tid)
                        for (int q = 0; q < 12; ++q) part[tid][q] += part[tid + N][q];
                if (need_A)
                    for (int q = 0; q < dim * dim; ++q) d_A[(size_t)n * dim * dim + q] += part[0][q];
                if (need_T)
                    for (int q = 0; q < dim; ++q) d_T[(size_t)n * dim + q] += part[0][9 + q];
            }
        }
    }
    free(part);
    return 0;
}

/* ---- regrid --------------------------------------------------------------- */

/* cuda/affine.cu:612-681 (kernels), :683-734 (host).  origin/spacing arrive as
 * double and are narrowed to REAL at the kernel call.  3D: hz starts at
 * Oz - oz*Sz and is advanced by `hz += Sz` per output k (affine.cu:669-675),
 * i.e. a sequentially rounded running sum -- restated as such. */
int FN(oracle_regrid_forward)(REAL *out, const REAL *I, int dim, long nn, long nc, long nx, long ny, long nz,
                              long Nx, long Ny, long Nz, const double *origin, const double *spacing) {
    if (dim != 2 && dim != 3) return -1;
    if (dim == 2) { nz = 1; Nz = 1; }
    const size_t nvox = (size_t)nx * ny * nz, Nvox = (size_t)Nx * Ny * Nz;
    const REAL Ox = (REAL)origin[0], Oy = (REAL)origin[1], Oz = dim == 3 ? (REAL)origin[2] : 0;
    const REAL Sx = (REAL)spacing[0], Sy = (REAL)spacing[1], Sz = dim == 3 ? (REAL)spacing[2] : 0;
    const REAL ox = (REAL)(.5 * (double)(REAL)(Nx - 1));
    const REAL oy = (REAL)(.5 * (double)(REAL)(Ny - 1));
    const REAL oz = (REAL)(.5 * (double)(REAL)(Nz - 1));
    for (long q = 0; q < nn * nc; ++q) {
        const REAL *In = I + (size_t)q * nvox;
        REAL *on = out + (size_t)q * Nvox;
        for (long i = 0; i < Nx; ++i)
            for (long j = 0; j < Ny; ++j) {
                REAL hx = LG_FMA((REAL)i - ox, Sx, Ox);
                REAL hy = LG_FMA((REAL)j - oy, Sy, Oy);
                if (dim == 2) {
                    on[(size_t)i * Ny + j] = FN(lg_bilerp)(In, hx, hy, nx, ny);
                } else {
                    REAL hz = LG_FMA(-oz, Sz, Oz);
                    for (long k = 0; k < Nz; ++k) {
                        on[((size_t)i * Ny + j) * Nz + k] = FN(lg_trilerp)(In, hx, hy, hz, nx, ny, nz);
                        hz += Sz;
                    }
                }
            }
    }
    return 0;
}

/* cuda/affine.cu:736-800 (kernels), :802-855 (host).  Here hz = (k-oz)*Sz+Oz
 * per voxel (affine.cu:791), not the running sum of the forward. */
int FN(oracle_regrid_backward)(REAL *d_I, const REAL *go, int dim, long nn, long nc, long nx, long ny, long nz,
                               long Nx, long Ny, long Nz, const double *origin, const double *spacing) {
    if (dim != 2 && dim != 3) return -1;
    if (dim == 2) { nz = 1; Nz = 1; }
    const size_t nvox = (size_t)nx * ny * nz, Nvox = (size_t)Nx * Ny * Nz;
    const REAL Ox = (REAL)origin[0], Oy = (REAL)origin[1], Oz = dim == 3 ? (REAL)origin[2] : 0;
    const REAL Sx = (REAL)spacing[0], Sy = (REAL)spacing[1], Sz = dim == 3 ? (REAL)spacing[2] : 0;
    const REAL ox = (REAL)(.5 * (double)(REAL)(Nx - 1));
    const REAL oy = (REAL)(.5 * (double)(REAL)(Ny - 1));
    const REAL oz = (REAL)(.5 * (double)(REAL)(Nz - 1));
    memset(d_I, 0, (size_t)nn * nc * nvox * sizeof(REAL));
    for (long q = 0; q < nn * nc; ++q) {
        REAL *dIn = d_I + (size_t)q * nvox;
        const REAL *gon = go + (size_t)q * Nvox;
        for (long i = 0; i < Nx; ++i)
            for (long j = 0; j < Ny; ++j) {
                REAL hx = LG_FMA((REAL)i - ox, Sx, Ox);
                REAL hy = LG_FMA((REAL)j - oy, Sy, Oy);
                if (dim == 2) {
                    FN(lg_splat2)(dIn, gon[(size_t)i * Ny + j], hx, hy, nx, ny);
                } else {
                    for (long k = 0; k < Nz; ++k) {
                        REAL hz = LG_FMA((REAL)k - oz, Sz, Oz);
                        FN(lg_splat3)(dIn, gon[((size_t)i * Ny + j) * Nz + k], hx, hy, hz, nx, ny, nz);
                    }
                }
            }
    }
    return 0;
}

#undef FN
#undef LG_CAT
#undef LG_CAT_
