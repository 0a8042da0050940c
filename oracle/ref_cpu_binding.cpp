// TEST INFRASTRUCTURE ONLY.
// Compiles the reference's own CPU path, /root/reference/lagomorph/extension/cpu/affine.cpp,
// from where it lies (no reference source is copied into this repo) and binds
// its one entry point, affine_interp_cpu_forward (cpu/affine.cpp:129-169), plus the
// reference's own interpolation cores biLerp / triLerp / biLerp_grad / triLerp_grad
// (include/interp.h:9-122 and :128-327 -- the header cpu/affine.cpp itself includes; it is
// host-compilable through its DEVICE macro, defs.h:44-48) evaluated at caller-given points.
// These are the functions every CUDA kernel of rows a1/a2/a9/a10 calls, so the oracle's
// restatement of them is pinned value by value against real reference code.
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

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("affine_interp_cpu_forward", &affine_interp_cpu_forward,
          "reference cpu/affine.cpp:129 affine_interp_cpu_forward");
    m.def("interp_points", &interp_points,
          "reference include/interp.h biLerp/triLerp (:9,:59) and biLerp_grad/triLerp_grad (:128,:206) at given points");
}
#endif
