// TEST INFRASTRUCTURE ONLY.
// Compiles the reference's own CPU path, /root/reference/lagomorph/extension/cpu/affine.cpp,
// from where it lies (no reference source is copied into this repo) and binds
// its one entry point, affine_interp_cpu_forward (cpu/affine.cpp:129-169), plus the
// reference's own interpolation cores biLerp / triLerp / biLerp_grad / triLerp_grad
// (include/interp.h:9-122 and :128-327 -- the header cpu/affine.cpp itself includes; it is
// host-compilable through its DEVICE macro, defs.h:44-48) evaluated at caller-given points.
// These are the functions every CUDA kernel of rows a1/a2/a9/a10 calls, so the oracle's
// restatement of them is pinned value by value against real reference code.  Since round 6 also the
// index rules of include/extrap.h (get_value_safe, clampBackground, map_point, isInside), see below.
//
// The reference was written against torch 1.0: it passes `Tensor::type()`
// (a DeprecatedTypeProperties) to AT_DISPATCH_FLOATING_TYPES, which torch 2.10
// only accepts as a ScalarType.  The macro is re-pointed at `.scalarType()`
// below; nothing else is adapted and the reference file is included unmodified.
#include <torch/extension.h>
#include <ATen/Dispatch.h>

#undef AT_DISPATCH_FLOATING_TYPES
#define AT_DISPATCH_FLOATING_TYPES(TYPE, NAME, ...) \
    AT_DISPATCH_SWITCH((TYPE).scalarType(), NAME, AT_DISPATCH_CASE_FLOATING_TYPES(__VA_ARGS__))

#ifndef LAGOMORPH_REF_NO_MODULE
bool lagomorph_debug_mode = false;  // extension.cpp:26 (declared extern in include/defs.h:15)
#endif

#include LAGOMORPH_REF_CPU_AFFINE

#ifndef LAGOMORPH_REF_NO_MODULE  // (the option-B build below links this file for affine_interp_cpu_forward only)

// img: (sx, sy[, sz]) contiguous; pts: (npts, dim) contiguous, same dtype.
// Returns (npts, 1 + dim): value, then the gradient -- from the reference's *_grad functions;
// column 0 of `lerp` is the reference's biLerp / triLerp value (the *_grad functions also return a
// value, Ix, which is returned in `grad`'s column 0).
template <typename Real>
static void ref_points(at::Tensor lerp, at::Tensor grad, at::Tensor img, at::Tensor pts) {
    const int dim = (int)img.dim();
    const Real *I = img.data_ptr<Real>();
    const Real *p = pts.data_ptr<Real>();
    Real *L = lerp.data_ptr<Real>();
    Real *G = grad.data_ptr<Real>();
    const int64_t n = pts.size(0);
    const int sx = (int)img.size(0), sy = (int)img.size(1), sz = dim == 3 ? (int)img.size(2) : 1;
    for (int64_t q = 0; q < n; ++q) {
        if (dim == 2) {
            L[q] = biLerp<Real, DEFAULT_BACKGROUND_STRATEGY>(I, p[2 * q], p[2 * q + 1], sx, sy);
            biLerp_grad<Real, DEFAULT_BACKGROUND_STRATEGY>(G[3 * q], G[3 * q + 1], G[3 * q + 2], I, p[2 * q],
                                                           p[2 * q + 1], sx, sy);
        } else {
            L[q] = triLerp<Real, DEFAULT_BACKGROUND_STRATEGY>(I, p[3 * q], p[3 * q + 1], p[3 * q + 2], sx, sy, sz);
            triLerp_grad<Real, DEFAULT_BACKGROUND_STRATEGY>(G[4 * q], G[4 * q + 1], G[4 * q + 2], G[4 * q + 3], I,
                                                            p[3 * q], p[3 * q + 1], p[3 * q + 2], sx, sy, sz);
        }
    }
}

static std::vector<at::Tensor> interp_points(at::Tensor img, at::Tensor pts) {
    TORCH_CHECK(img.is_contiguous() && pts.is_contiguous() && img.scalar_type() == pts.scalar_type());
    TORCH_CHECK((img.dim() == 2 || img.dim() == 3) && pts.dim() == 2 && pts.size(1) == img.dim());
    auto lerp = at::empty({pts.size(0)}, img.options());
    auto grad = at::empty({pts.size(0), 1 + img.dim()}, img.options());
    if (img.scalar_type() == at::kFloat)
        ref_points<float>(lerp, grad, img, pts);
    else
        ref_points<double>(lerp, grad, img, pts);
    return {lerp, grad};
}

// ---- include/extrap.h (pulled in by include/interp.h; every function there is spelled with the DEVICE macro, so it
// host-compiles as it lies): the index rules the kernels of rows a2, a4-a7, a9, a10 rest on.
//  * get_value_safe<Real, CLAMP> (extrap.h:110-192): the accessor every diff_x / diff_y / diff_z of include/diff.h:7-52
//    calls.  diff.h itself spells `__device__` literally and cannot be host-compiled without a stand-in, so its one-line
//    formula 0.5f * (f(i+1) - f(i-1)) is written out HERE around the reference's own accessor (`central_differences`).
//  * clampBackground (extrap.h:46-77), map_point<CLAMP> (extrap.h:194-253), isInside (extrap.h:24-38): the clamped
//    (floor, ceil) index pairs of the interpolation footprint and of the splat's target cells.
template <typename Real>
static void ref_extrap(at::Tensor val, at::Tensor grad, at::Tensor arr, at::Tensor idx) {
    const int dim = (int)arr.dim();
    const Real *a = arr.data_ptr<Real>();
    const int64_t *q = idx.data_ptr<int64_t>();
    Real *v = val.data_ptr<Real>();
    Real *g = grad.data_ptr<Real>();
    const int nx = (int)arr.size(0), ny = (int)arr.size(1), nz = dim == 3 ? (int)arr.size(2) : 1;
    constexpr auto S = BACKGROUND_STRATEGY_CLAMP;
    for (int64_t n = 0; n < idx.size(0); ++n) {
        if (dim == 2) {
            const int i = (int)q[2 * n], j = (int)q[2 * n + 1];
            v[n] = get_value_safe<Real, S>(a, (size_t)nx, (size_t)ny, i, j);
            g[2 * n] = 0.5f * (get_value_safe<Real, S>(a, (size_t)nx, (size_t)ny, i + 1, j) - get_value_safe<Real, S>(a, (size_t)nx, (size_t)ny, i - 1, j));      // diff.h:13-14
            g[2 * n + 1] = 0.5f * (get_value_safe<Real, S>(a, (size_t)nx, (size_t)ny, i, j + 1) - get_value_safe<Real, S>(a, (size_t)nx, (size_t)ny, i, j - 1));  // diff.h:22-23
        } else {
            const int i = (int)q[3 * n], j = (int)q[3 * n + 1], k = (int)q[3 * n + 2];
            v[n] = get_value_safe<Real, S>(a, nx, ny, nz, i, j, k);
            g[3 * n] = 0.5f * (get_value_safe<Real, S>(a, nx, ny, nz, i + 1, j, k) - get_value_safe<Real, S>(a, nx, ny, nz, i - 1, j, k));      // diff.h:32-33
            g[3 * n + 1] = 0.5f * (get_value_safe<Real, S>(a, nx, ny, nz, i, j + 1, k) - get_value_safe<Real, S>(a, nx, ny, nz, i, j - 1, k));  // diff.h:41-42
            g[3 * n + 2] = 0.5f * (get_value_safe<Real, S>(a, nx, ny, nz, i, j, k + 1) - get_value_safe<Real, S>(a, nx, ny, nz, i, j, k - 1));  // diff.h:50-51
        }
    }
}

// arr: (nx, ny[, nz]) contiguous; idx: (npts, dim) int64 (any integers).  Returns the accessor's value (npts) and the
// clamped central differences (npts, dim).
static std::vector<at::Tensor> extrap_points(at::Tensor arr, at::Tensor idx) {
    TORCH_CHECK(arr.is_contiguous() && idx.is_contiguous() && idx.scalar_type() == at::kLong);
    TORCH_CHECK((arr.dim() == 2 || arr.dim() == 3) && idx.dim() == 2 && idx.size(1) == arr.dim());
    auto val = at::empty({idx.size(0)}, arr.options());
    auto grad = at::empty({idx.size(0), arr.dim()}, arr.options());
    if (arr.scalar_type() == at::kFloat)
        ref_extrap<float>(val, grad, arr, idx);
    else
        ref_extrap<double>(val, grad, arr, idx);
    return {val, grad};
}

// fc: (npts, 2 * dim) int64 rows (floorX, floorY[, floorZ], ceilX, ceilY[, ceilZ]); sizes: dim extents.
// Returns (npts, 2 * dim + 2): the indices after map_point<CLAMP>, its return value, and isInside of the UNMAPPED indices.
static at::Tensor map_points(at::Tensor fc, std::vector<int64_t> sizes) {
    TORCH_CHECK(fc.is_contiguous() && fc.scalar_type() == at::kLong && fc.dim() == 2);
    const int dim = (int)sizes.size();
    TORCH_CHECK((dim == 2 || dim == 3) && fc.size(1) == 2 * dim);
    auto out = at::empty({fc.size(0), 2 * dim + 2}, fc.options());
    const int64_t *q = fc.data_ptr<int64_t>();
    int64_t *o = out.data_ptr<int64_t>();
    for (int64_t n = 0; n < fc.size(0); ++n) {
        const int64_t *r = q + n * 2 * dim;
        int64_t *w = o + n * (2 * dim + 2);
        if (dim == 2) {
            int fx = (int)r[0], fy = (int)r[1], cx = (int)r[2], cy = (int)r[3];
            w[5] = isInside(fx, fy, cx, cy, (int)sizes[0], (int)sizes[1]) ? 1 : 0;
            w[4] = map_point<BACKGROUND_STRATEGY_CLAMP>(fx, fy, cx, cy, (size_t)sizes[0], (size_t)sizes[1]) ? 1 : 0;
            w[0] = fx; w[1] = fy; w[2] = cx; w[3] = cy;
        } else {
            int fx = (int)r[0], fy = (int)r[1], fz = (int)r[2], cx = (int)r[3], cy = (int)r[4], cz = (int)r[5];
            w[7] = isInside(fx, fy, fz, cx, cy, cz, (int)sizes[0], (int)sizes[1], (int)sizes[2]) ? 1 : 0;
            w[6] = map_point<BACKGROUND_STRATEGY_CLAMP>(fx, fy, fz, cx, cy, cz, (size_t)sizes[0], (size_t)sizes[1], (size_t)sizes[2]) ? 1 : 0;
            w[0] = fx; w[1] = fy; w[2] = fz; w[3] = cx; w[4] = cy; w[5] = cz;
        }
    }
    return out;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("extrap_points", &extrap_points,
          "reference include/extrap.h get_value_safe<CLAMP> (:110-192) at integer indices, and the clamped central differences of include/diff.h:7-52 spelled out around it");
    m.def("map_points", &map_points,
          "reference include/extrap.h map_point<CLAMP> (:194-253, via clampBackground :46-77) and isInside (:24-38) on (floor, ceil) index rows");
    m.def("affine_interp_cpu_forward", &affine_interp_cpu_forward,
          "reference cpu/affine.cpp:129 affine_interp_cpu_forward");
    m.def("interp_points", &interp_points,
          "reference include/interp.h biLerp/triLerp (:9,:59) and biLerp_grad/triLerp_grad (:128,:206) at given points");
}
#endif
