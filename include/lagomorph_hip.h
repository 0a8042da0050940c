/*
 * lagomorph_hip.h -- C ABI of liblagomorph_hip.so, the MI355X (gfx950) HIP
 * implementation of lagomorph's deform / interp / metric / adjrep hot path.
 *
 * This is the drop-in boundary: each entry point replaces one function of the
 * reference's `lagomorph_ext` pybind11 module
 * (/root/reference/lagomorph/extension/extension.cpp:175-189).  Signatures are
 * plain C: raw device pointers, 64-bit extents, scalar parameters, flags, and
 * the hipStream_t to launch on (as void*; NULL = the null stream).  No torch
 * types, no allocation, no host synchronisation (except in debug mode).
 *
 * Conventions
 *  - Every tensor is dense, contiguous, batch-major N C (D) H W with the LAST
 *    spatial axis fastest, exactly as the reference indexes it
 *    (include/extrap.h:15-21: index = (x*sizeY + y)*sizeZ + z).
 *  - `dim` is 2 or 3.  For dim == 2 pass nz = 1 (ignored).
 *  - Each function exists for float (`_f32`) and double (`_f64`); the
 *    reference dispatches both via AT_DISPATCH_FLOATING_TYPES.
 *  - Outputs are written in full by the call (splat targets are zeroed on the
 *    stream before the kernel), so callers may pass uninitialised buffers.
 *  - Return value: LAGO_OK (0), LAGO_ERR_INVALID (-1: bad argument; nothing
 *    was launched) or LAGO_ERR_HIP (-2: a HIP runtime call failed).
 *    lago_last_error() returns a thread-local description.
 *  - Extent limits: nx*ny*nz < 2^29 per batch item (one channel plane is addressed
 *    with 32-bit byte offsets); whole tensors are addressed with 64-bit offsets.
 */
#ifndef LAGOMORPH_HIP_H
#define LAGOMORPH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LAGO_OK 0
#define LAGO_ERR_INVALID (-1)
#define LAGO_ERR_HIP (-2)

#define LAGO_ABI_VERSION 5

/* ---- housekeeping --------------------------------------------------------- */

/* replaces set_debug_mode (extension.cpp:105-107): when non-zero every entry
 * point synchronises its stream after the launch and reports kernel faults
 * through its return value (the reference only printed them, defs.h:17-23). */
void lago_set_debug(int on);
int lago_get_debug(void);
int lago_abi_version(void);
const char *lago_version(void);
const char *lago_last_error(void);

/* Tuning settings: ONE struct, read and written as a whole (the ABI does not grow per experiment).  The settings are
 * PROCESS-WIDE (two users of the library in one process share them), stored atomically field by field (they may be
 * changed while other host threads, e.g. autograd's backward threads, are inside entry points) and affect speed only,
 * never which results are produced beyond the rounding differences noted per field.  The defaults are what the product
 * runs with; the parity tests sweep them.  `struct_size` is sizeof(lago_tuning) as the CALLER was compiled: the
 * library reads / writes only that many bytes, so a caller built against an older header keeps working when fields
 * are appended. */
typedef struct lago_tuning {
    uint32_t struct_size;
    /* interp_backward: 0 global float atomics only (the reference's form), 1 LDS-privatised splat with atomic flush
     * (default for 3D float32) */
    int32_t splat_mode;
    /* general tiled LDS splat: source tile TX TY TZ (TZ = 0: whole z rows, split evenly above 128 voxels; TX = 0: about
     * 4096 voxels per tile), window margins MX MY MZ, threads per workgroup (256 / 512 / 1024).  Default 0 8 0 1 1 4 512 */
    int32_t splat_tile[7];
    /* sheared-window float32 3D displacement splat (csrc/splat.hip: splat_shear_kernel, the form interp_backward takes
     * for float32 3D fields): on (0 leaves every call to the general tiled kernel), source tile TX TY TZ (TX an upper
     * bound shrunk until the window fits 80 KB; TZ = 0: whole z rows, split evenly above 128 voxels), window margins
     * MX MY MZ around the probed origin, threads per workgroup.  Default 1, 8 6 0, 1 1 4, 1024.  d_u is bit-identical
     * under every setting; d_I differs by the order of its sums */
    int32_t splat_shear[8];
    /* several channels with d_u wanted, sheared-window kernel: 2 (default) each voxel's geometry -- window addresses,
     * gather offset, fractions -- and its d_u sums stay in registers over the channel loop (tiles of at most 2048
     * voxels), 1 only the d_u sums do (geometry recomputed per channel), 0 d_u is read-modify-written per channel.
     * Same d_u bits under every setting */
    int32_t splat_shear_mc;
    /* the same for the general tiled kernel: 1 (default) / 0 */
    int32_t splat_mc;
    /* 1 (default): the slab-unrolled 3D gather kernels (two voxels per lane) when the shape allows; 0: one-voxel-per-lane
     * kernels only */
    int32_t vector_kernels;
    /* 1 (default): every launch of the large kernels walks its workgroups in the opposite direction of the launch before
     * it, so that a consumer starts on what its producer wrote last -- still in the 256 MB Infinity Cache; 0: always
     * ascending.  Same results (scatter-add outputs differ in their last bits, as between any two runs).  The direction
     * is taken from ONE process-wide launch counter (every entry point that builds a launch geometry advances it, small
     * launches and calls that fail a later argument check included), so "opposite to its producer" holds for the chains
     * of large kernels the library itself issues back to back, and the last bits of scatter-add outputs (d_I) depend on
     * how many launches came before -- as they depend on the hardware's atomic ordering anyway; every other output is
     * bit-identical in both directions */
    int32_t launch_order;
    /* 1 (default): the 3D Jacobian / stencil terms of Ad_star and jacobian_times_vectorfield_backward are taken from an
     * LDS-staged tile of z-rows with a one-voxel halo (csrc/stencil_tile.hpp) where the shape allows -- 128^3 and 160^3
     * volumes through instantiations with their geometry compiled in; 3: row tiles without those instantiations; 0:
     * every neighbour is loaded from global memory.  Same bits */
    int32_t stencil_tile;
    /* 1 (default): float32 3D trilinear gathers of smooth fields (compose, interp_forward of several channels) stage the
     * source block of a tile of voxels in LDS with LDS-direct loads and take the corners from there
     * (csrc/gather_window.hpp) where the shape allows; 0: pair gathers through the vector L1 only.  Same bits */
    int32_t gather_window;
    /* lago_fluid_metric implementation.  3 (default): the tuned LDS-tiled FFT passes where they apply (float32 3D with nx
     * in {64,96,128,160,192,256} and (ny, nz) any pair of {64,96,128,160,192} or one of the power-of-two planes
     * 32x{64,128,256}, 64x256, 128x256, 256x{64,128}: lengths 2^a, 3*2^a, 5*2^a; since round 6 also nx in {176,208} and
     * the planes 208x176, 176x176, 176x208 -- 11*16 and 13*16: the 176 x 208 x 176 brain grid -- and nx in {112,144,224,240}
     * with the planes 112x96, 96x112, 112x112, 128x112, 112x128, 224x160, 160x224, 224x128, 144x144, 176x144, 144x176,
     * 240x160, 160x240 (odd factors 7, 9, 15), and nx in {80,88,104,120} with the planes 80x80, 104x88, 88x88, 88x104, 120x120
     * (a plane with ny % 16 = 8 needs nx % 16 = 8 as well); planes above the LDS or not in these lists, ny in {128,144,...,256}
     * (multiples of 16 with an odd factor up to 15) and nz from the same set -- since the lists hold every multiple of 16
     * from 64 to 256, ANY volume with such extents -- as rows + columns (five launches:
     * 256^3, 160 x 192 x 224, 192 x 224 x 192 ...); float32 2D planes up to 128 x 128 in
     * one fused kernel) and the generic hand-written passes of csrc/fftg.hip for every other shape and for float64
     * (any extent up to 4096 (float32) / 2048 (float64) points per axis, lines with a prime factor >= 29 through
     * Bluestein's convolution; a longer axis is the one case left to the spot-checked rocFFT path): no rocFFT call
     * otherwise.  2: the tuned passes, rocFFT for the rest (1: rocFFT 2D
     * (y, z) plan + fused x-axis pass, nx in {64,128,256}; 0: rocFFT 3D plan + operator kernel; a mode falls back to
     * the next lower one for shapes it does not support); the rocFFT plans are spot-checked against a direct DFT
     * (csrc/fft.hip).  Results agree to rounding.  4: as 3, with the generic passes' x transforms and operator as three
     * launches instead of the fused one (csrc/fftg.hip: fft_xop_kernel) -- the same bits; the comparison switch */
    int32_t fluid_mode;
    /* FFT-pass fluid metric: batch items per x-pass workgroup (0, the default: chosen by the size of the launch) */
    int32_t fluid_xpass_ipw;
    /* persistent prefetching zy kernels for planes above 80 KB of LDS (default 1) */
    int32_t fluid_zy_persist;
    /* 512-thread x-pass workgroups for the 256-point tile (default 1) */
    int32_t fluid_xpass_wide;
    /* x pass as a persistent grid of two workgroups per CU that prefetch their next tile (default 1: taken when the
     * launch has at least four (bin tile, batch item) pairs per workgroup; 2: always; 0: never).  Same bits */
    int32_t fluid_xpass_persist;
    /* affine_interp_backward (3D), the image splat: 1 (default) by target boxes for batch items whose matrix is
     * invertible with a moderate inverse (decided per item on the device), the general tiled LDS splat for the others;
     * 0: the general tiled splat for every item.  d_I differs by the order of its sums */
    int32_t affine_box;
} lago_tuning;
/* the settings in force / the library's defaults: fills t->struct_size bytes (struct_size set by the caller) */
void lago_get_tuning(lago_tuning *t);
void lago_default_tuning(lago_tuning *t);
/* applies the first t->struct_size bytes (fields beyond them keep their values); LAGO_ERR_INVALID for a null pointer or
 * a struct_size that is not a whole number of fields */
int lago_set_tuning(const lago_tuning *t);
/* Which implementation calls were dispatched to: number of launches so far in this process per path (telemetry; the
 * tests use it to make sure a case meant to exercise a fast path really runs it).  -1 for an unknown id. */
#define LAGO_PATH_GATHER_WINDOW 0  /* compose through the LDS window */
#define LAGO_PATH_STENCIL_TILE 1   /* Ad_star row-tile kernel */
#define LAGO_PATH_VECTOR_GATHER 2  /* slab-unrolled 3D gather kernels (interp_forward, compose, Ad_star) */
#define LAGO_PATH_SPLAT_SHEAR 3    /* sheared-window splat */
#define LAGO_PATH_SPLAT_SHEAR_MC 4 /* its geometry-once multi-channel form */
#define LAGO_PATH_SPLAT_TILED 5    /* tiled LDS splat */
#define LAGO_PATH_SPLAT_GLOBAL 6   /* global-atomics splat (the reference's form) */
#define LAGO_PATH_FLUID_LDS 7      /* lago_fluid_metric: three hand-written FFT passes */
#define LAGO_PATH_FLUID_2D 8       /* lago_fluid_metric: one fused 2D kernel */
#define LAGO_PATH_FLUID_XPASS 9    /* lago_fluid_metric: rocFFT (y, z) plan + fused x pass */
#define LAGO_PATH_FLUID_ROCFFT 10  /* lago_fluid_metric: rocFFT plan + operator kernel */
#define LAGO_PATH_SPLAT_2D 11      /* LDS-privatised 2D splat (interp_backward of 2D fields) */
#define LAGO_PATH_SPLAT_AFFINE_BOX 12 /* affine_interp_backward's image splat by target boxes */
#define LAGO_PATH_FLUID_GENERIC 13 /* lago_fluid_metric: generic hand-written FFT passes (any extent, both precisions) */
long long lago_path_launches(int path);
/* Launch direction (lago_tuning.launch_order): launches so far that walked their workgroups in DESCENDING order.  Kernels
 * whose results do not depend on the block order alternate (a consumer starts on what the Infinity Cache still holds of
 * its producer's output); every launch that feeds a scatter-add (interp_backward, interp_hessian_diagonal_image,
 * affine_interp_backward, regrid_backward) walks ascending and leaves the alternation alone, so that the arrival order
 * of its float atomics -- the last bits of d_I, d_A, d_T -- never depends on the calls made before it.  Telemetry for
 * tests/test_gpu_dispatch.py::test_scatter_direction_is_history_free. */
long long lago_reversed_launches(void);


#define LAGO_DECLARE(REAL, SUF)                                                                                      \
    /* interp_forward (extension.cpp:135-143 -> cuda/interp.cu:80-130):                                           \
     * out[n,c,x] = lerp(I[n or 0,c], x + dt*u[n,:,x]), clamp boundary.                                             \
     * I: (broadcast_I ? 1 : nn, nc, sp)  u: (nn, dim, sp)  out: (nn, nc, sp) */                                    \
    int lago_interp_forward##SUF(REAL *out, const REAL *I, const REAL *u, double dt, int dim, int64_t nn,           \
                                 int64_t nc, int64_t nx, int64_t ny, int64_t nz, int broadcast_I, void *stream);    \
    /* interp_backward (extension.cpp:145-156 -> cuda/interp.cu:246-313): d_I like I (splat of grad_out),           \
     * d_u like u.  Both are always produced; a gradient that is not needed is all zeros, as in the reference. */   \
    int lago_interp_backward##SUF(REAL *d_I, REAL *d_u, const REAL *grad_out, const REAL *I, const REAL *u,         \
                                  double dt, int dim, int64_t nn, int64_t nc, int64_t nx, int64_t ny, int64_t nz,   \
                                  int broadcast_I, int need_I, int need_u, void *stream);                           \
    /* interp_hessian_diagonal_image (cuda/interp.cu:351-381), 2D only.  out like I: (nI, nc, nx, ny);              \
     * as in the reference every (n, c) accumulates into plane 0 of out. */                                         \
    int lago_interp_hessian_diagonal_image##SUF(REAL *out, const REAL *u, double dt, int64_t nI, int64_t nn,        \
                                                int64_t nc, int64_t nx, int64_t ny, void *stream);                  \
    /* jacobian_times_vectorfield_forward (cuda/diff.cu:129-185).  v: (nn, nc, sp) is differentiated,               \
     * w: (nn, dim, sp) is contracted; out like v.  displacement/transpose require nc == dim. */                    \
    int lago_jtv_forward##SUF(REAL *out, const REAL *v, const REAL *w, int displacement, int transpose, int dim,    \
                              int64_t nn, int64_t nc, int64_t nx, int64_t ny, int64_t nz, void *stream);            \
    /* jacobian_times_vectorfield_backward (cuda/diff.cu:475-540): d_v like v, d_w like w, both always. */          \
    int lago_jtv_backward##SUF(REAL *d_v, REAL *d_w, const REAL *grad_out, const REAL *v, const REAL *w,            \
                               int displacement, int transpose, int dim, int64_t nn, int64_t nc, int64_t nx,        \
                               int64_t ny, int64_t nz, void *stream);                                               \
    /* jacobian_times_vectorfield_adjoint_forward (cuda/diff.cu:634-672): out[c] = sum_d D_d^T (w_d z_c).           \
     * z: (nn, nc, sp), w: (nn, dim, sp), out like z. */                                                            \
    int lago_jtv_adjoint_forward##SUF(REAL *out, const REAL *z, const REAL *w, int dim, int64_t nn, int64_t nc,     \
                                      int64_t nx, int64_t ny, int64_t nz, void *stream);                            \
    /* jacobian_times_vectorfield_adjoint_backward (cuda/diff.cu:783-835), nc == dim. */                            \
    int lago_jtv_adjoint_backward##SUF(REAL *d_v, REAL *d_w, const REAL *grad_out, const REAL *v, const REAL *w,    \
                                       int dim, int64_t nn, int64_t nx, int64_t ny, int64_t nz, void *stream);      \
    /* fluid_operator (extension.cpp:158-173 -> cuda/metric.cu:308-355): in place on the interleaved-complex        \
     * rFFT buffer Fm of shape (nn, dim, nx, ny[, nz], 2) where the last spatial extent is the half-spectrum        \
     * length; cos/sin LUTs have nx, ny, nz entries (cosZ/sinZ unused for dim == 2). */                             \
    int lago_fluid_operator##SUF(REAL *Fm, int inverse, const REAL *cosX, const REAL *sinX, const REAL *cosY,       \
                                 const REAL *sinY, const REAL *cosZ, const REAL *sinZ, double alpha, double beta,   \
                                 double gamma, int dim, int64_t nn, int64_t nx, int64_t ny, int64_t nz,             \
                                 void *stream);                                                                     \
    /* affine_interp_forward (extension.cpp:109-118 -> cuda/affine.cu:114-169).                                     \
     * I: (broadcast_I ? 1 : nn, nc, sp)  A: (nn, dim, dim)  T: (nn, dim)  out: (nn, nc, sp) */                     \
    int lago_affine_interp_forward##SUF(REAL *out, const REAL *I, const REAL *A, const REAL *T, int dim,            \
                                        int64_t nn, int64_t nc, int64_t nx, int64_t ny, int64_t nz,                 \
                                        int broadcast_I, void *stream);                                             \
    /* affine_interp_backward (extension.cpp:120-133 -> cuda/affine.cu:538-610).  Outputs that are not needed       \
     * may be NULL (the reference returns size-0 tensors for them). */                                              \
    int lago_affine_interp_backward##SUF(REAL *d_I, REAL *d_A, REAL *d_T, const REAL *grad_out, const REAL *I,      \
                                         const REAL *A, const REAL *T, int dim, int64_t nn, int64_t nc,             \
                                         int64_t nx, int64_t ny, int64_t nz, int broadcast_I, int need_I,           \
                                         int need_A, int need_T, void *stream);                                     \
    /* regrid_forward (cuda/affine.cu:683-734): resample I (nn, nc, nx, ny, nz) onto an (Nx, Ny, Nz) grid with      \
     * the given origin/spacing (length-dim host arrays of doubles). */                                             \
    int lago_regrid_forward##SUF(REAL *out, const REAL *I, int dim, int64_t nn, int64_t nc, int64_t nx,             \
                                 int64_t ny, int64_t nz, int64_t Nx, int64_t Ny, int64_t Nz, const double *origin,  \
                                 const double *spacing, void *stream);                                              \
    /* regrid_backward (cuda/affine.cu:802-855): splat grad_out (nn, nc, Nx, Ny, Nz) back onto (nx, ny, nz). */     \
    int lago_regrid_backward##SUF(REAL *d_I, const REAL *grad_out, int dim, int64_t nn, int64_t nc, int64_t nx,     \
                                  int64_t ny, int64_t nz, int64_t Nx, int64_t Ny, int64_t Nz,                       \
                                  const double *origin, const double *spacing, void *stream);                       \
    /* the same operator applied axis by axis in gather form (positive spacings): no atomics, d_I need not be       \
     * initialised, results independent of the launch; only the association of the weight products differs from     \
     * the reference's (wz (wy (wx m)) against wx wy wz m).  ws: two temporaries of ws_elems / 2 elements each,     \
     * ws_elems >= 2 nn nc max(nx Ny Nz, nx ny Nz) (3D; 2D: >= 2 nn nc nx Ny with (nx, ny) the 2D extents; may be   \
     * null when ws_elems == 0 suffices).  Inputs the passes do not cover (a non-positive spacing, spacings        \
     * >= 1e6 or so small that rounding moves a sample by more than 64 output cells, |origin| >= 1e9, 2^31 elements  \
     * or more in a pass) are served by lago_regrid_backward: every call the reference accepts is accepted.          \
     * Not in the reference's extension surface. */                                                                 \
    int lago_regrid_backward_sep##SUF(REAL *d_I, const REAL *grad_out, REAL *ws, int64_t ws_elems, int dim,         \
                                      int64_t nn, int64_t nc, int64_t nx, int64_t ny, int64_t nz, int64_t Nx,       \
                                      int64_t Ny, int64_t Nz, const double *origin, const double *spacing,          \
                                      void *stream);

LAGO_DECLARE(float, _f32)
LAGO_DECLARE(double, _f64)

/* ---- fused geometry operators (beyond the reference's extension surface) -------------
 * compose: out = ds*u + dt*interp(v, u, ds) in one pass -- deform.compose
 * (/root/reference/lagomorph/deform.py:53-55), which the reference evaluates as one interp
 * kernel plus three elementwise kernels.  u, v, out: (nn, dim, sp).  Bit-identical to the
 * unfused expression (the three roundings are kept). */
int lago_compose_f32(float *out, const float *u, const float *v, double ds, double dt, int dim, int64_t nn,
                     int64_t nx, int64_t ny, int64_t nz, void *stream);
int lago_compose_f64(double *out, const double *u, const double *v, double ds, double dt, int dim, int64_t nn,
                     int64_t nx, int64_t ny, int64_t nz, void *stream);

/* lincomb: out[i] = c0 x0[i] + c1 x1[i] + ... (k = 1..4 terms, n elements), evaluated left to right with one fma per
 * term in the tensors' precision -- the elementwise sums of lddmm_step (lddmm.py:300-325: the regulariser's gradient
 * joining the velocity gradient, and `m.add_(-lr, p)` with p the sum of three gradient contributions) as one pass
 * each instead of torch's chain of mul / add kernels.  out may alias any input; unused inputs may be NULL. */
int lago_lincomb_f32(float *out, int k, const float *x0, const float *x1, const float *x2, const float *x3, double c0,
                     double c1, double c2, double c3, int64_t n, void *stream);
int lago_lincomb_f64(double *out, int k, const double *x0, const double *x1, const double *x2, const double *x3,
                     double c0, double c1, double c2, double c3, int64_t n, void *stream);

/* Ad_star (the coadjoint action of a diffeomorphism): out = (D phiinv + I) (m o (id + phiinv)) -- adjrep.Ad_star
 * (/root/reference/lagomorph/adjrep.py:86-97), which the reference evaluates as interp_forward
 * followed by jacobian_times_vectorfield_forward(displacement = true).  phiinv, m, out:
 * (nn, dim, sp); out may not alias an input.  Bit-identical to the two-call sequence (the
 * resampled momentum is rounded where that sequence stores it).  mphi (nullable, like m): when given, the
 * resampled momentum m o (id + phiinv) -- the interp_forward output of the two-call sequence -- is stored
 * there as well, so that a backward pass need not recompute it. */
int lago_Ad_star_f32(float *out, float *mphi, const float *phiinv, const float *m, int dim, int64_t nn, int64_t nx,
                     int64_t ny, int64_t nz, void *stream);
int lago_Ad_star_f64(double *out, double *mphi, const double *phiinv, const double *m, int dim, int64_t nn,
                     int64_t nx, int64_t ny, int64_t nz, void *stream);

/* interp_backward with start values (the fused backward forms of compose, Ad_star and expmap).  i_mode 0: d_I is
 * zeroed and receives the splat, exactly as lago_interp_backward; i_mode 1 (needs need_I): the splat is ADDED onto
 * the contents of d_I (a gradient accumulated over several calls, e.g. d/d m0 over the Euler steps of expmap).  The reference kernel owns d_u[n, d, x] in one thread and sums the channels' terms in
 * ascending order starting from zero (cuda/interp.cu:185-244); here the sum starts from
 *   u_mode 0: zero (identical to lago_interp_backward with need_u),
 *   u_mode 1: the contents of d_u on entry (d_u += ...: the `d_v.add_(d_u)` of a chain rule, without the extra pass),
 *   u_mode 2: addgo * grad_out[n, d, x] (needs nc == dim: the `ds * grad` term of compose's backward).
 * d_u is always produced (need_u is implied). */
int lago_interp_backward_fused_f32(float *d_I, float *d_u, const float *grad_out, const float *I, const float *u,
                                   double dt, int dim, int64_t nn, int64_t nc, int64_t nx, int64_t ny, int64_t nz,
                                   int broadcast_I, int need_I, int i_mode, int u_mode, double addgo, void *stream);
int lago_interp_backward_fused_f64(double *d_I, double *d_u, const double *grad_out, const double *I, const double *u,
                                   double dt, int dim, int64_t nn, int64_t nc, int64_t nx, int64_t ny, int64_t nz,
                                   int broadcast_I, int need_I, int i_mode, int u_mode, double addgo, void *stream);

/* jacobian_times_vectorfield_backward (lago_jtv_backward) whose d_v is added onto the contents of d_v when acc_v is
 * non-zero (d_w is always overwritten): the chain-rule sum of expmap's reverse sweep without a separate add pass. */
int lago_jtv_backward_acc_f32(float *d_v, float *d_w, const float *grad_out, const float *v, const float *w,
                              int displacement, int transpose, int dim, int64_t nn, int64_t nc, int64_t nx, int64_t ny,
                              int64_t nz, int acc_v, void *stream);
int lago_jtv_backward_acc_f64(double *d_v, double *d_w, const double *grad_out, const double *v, const double *w,
                              int displacement, int transpose, int dim, int64_t nn, int64_t nc, int64_t nx, int64_t ny,
                              int64_t nz, int acc_v, void *stream);

/* ad_star (the infinitesimal coadjoint action): out = (Dv)^T m - sum_d D_d^T (v_d m) -- adjrep.ad_star
 * (/root/reference/lagomorph/adjrep.py:69-83), which the reference evaluates as
 * jacobian_times_vectorfield_forward(v, m, transpose = true) minus
 * jacobian_times_vectorfield_adjoint_forward(m, v).  v, m, out: (nn, dim, sp); out may not alias an input.
 * Bit-identical to the three-call sequence. */
int lago_ad_star_f32(float *out, const float *v, const float *m, int dim, int64_t nn, int64_t nx, int64_t ny,
                     int64_t nz, void *stream);
int lago_ad_star_f64(double *out, const double *v, const double *m, int dim, int64_t nn, int64_t nx, int64_t ny,
                     int64_t nz, void *stream);

/* fluid_metric: the whole FluidMetricOperator.forward of the reference
 * (/root/reference/lagomorph/metric.py:11-19) in one call: out = irfft(L^(+-2) rfft(m)).
 * m, out: (nn, dim, nx, ny[, nz]) real; work: caller-provided scratch for the half spectrum,
 * nn*dim*nx*ny*(nz/2+1)*2 reals for dim == 3 (nn*dim*nx*(ny/2+1)*2 for dim == 2), clobbered.
 * m is not modified; out may not alias m.  LUTs as for lago_fluid_operator.
 * lut_generation: the float32 3D fast paths tabulate the per-frequency coefficients once per
 * (lut_generation, shape, alpha/beta/gamma, direction) in a library-owned device buffer (24 bytes per
 * frequency bin; the cache is bounded to 1 GiB and evicts oldest-first).  The table is a function of the
 * LUT *contents*, so the caller names a set of contents by a non-zero generation number and passes a new
 * number whenever it passes different contents (another shape, a refilled buffer); pointers are not part of
 * the key.  0 means "do not cache": the call takes the table-free path (rocFFT 3D plan + operator kernel).
 * hipFFT plans are cached inside the library per shape; their SetStream + Exec pairs are serialised. */
int lago_fluid_metric_f32(float *out, const float *m, float *work, int64_t lut_generation, int inverse,
                          const float *cosX, const float *sinX, const float *cosY, const float *sinY,
                          const float *cosZ, const float *sinZ, double alpha, double beta, double gamma, int dim,
                          int64_t nn, int64_t nx, int64_t ny, int64_t nz, void *stream);
int lago_fluid_metric_f64(double *out, const double *m, double *work, int64_t lut_generation, int inverse,
                          const double *cosX, const double *sinX, const double *cosY, const double *sinY,
                          const double *cosZ, const double *sinZ, double alpha, double beta, double gamma, int dim,
                          int64_t nn, int64_t nx, int64_t ny, int64_t nz, void *stream);
/* out = out_scale * (the field lago_fluid_metric returns): the factor multiplies the finished value in the field's
 * precision, i.e. the bits of a separate `out *= out_scale` pass, without that pass where the last kernel can take the
 * factor (the tuned 3D passes, the fused 2D kernel); one in-place pass inside the call otherwise.  For the Euler step
 * from the identity, phi_1 = -dt * sharp(m0) (lddmm.py:39-44 with a zero displacement; lagomorph_amd/lddmm.py). */
int lago_fluid_metric_scaled_f32(float *out, const float *m, float *work, int64_t lut_generation, int inverse,
                                 const float *cosX, const float *sinX, const float *cosY, const float *sinY,
                                 const float *cosZ, const float *sinZ, double alpha, double beta, double gamma, int dim,
                                 int64_t nn, int64_t nx, int64_t ny, int64_t nz, double out_scale, void *stream);
int lago_fluid_metric_scaled_f64(double *out, const double *m, double *work, int64_t lut_generation, int inverse,
                                 const double *cosX, const double *sinX, const double *cosY, const double *sinY,
                                 const double *cosZ, const double *sinZ, double alpha, double beta, double gamma, int dim,
                                 int64_t nn, int64_t nx, int64_t ny, int64_t nz, double out_scale, void *stream);
/* Drops every cached coefficient table (buffers still read by enqueued kernels are freed after them). */
void lago_fluid_cache_clear(void);
int lago_fluid_cache_entries(void);
/* rocFFT fallback plans (float64, 2D planes beyond the LDS, extents outside 2^a / 3*2^a / 5*2^a): how many are cached
 * and how many of them are currently verified by the spot check against a direct DFT (csrc/fft.hip: creating a plan
 * marks every plan unverified again; a check on an all-zero field or during stream capture proves nothing and leaves
 * the plan unverified).  Telemetry for the tests. */
int lago_fft_plan_state(int *plans, int *verified);

#ifdef __cplusplus
}
#endif
#endif /* LAGOMORPH_HIP_H */
