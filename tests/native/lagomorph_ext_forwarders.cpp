// INTEGRATION.md option B, written out: the 12 host functions that lagomorph's extension.cpp forward-declares
// (/root/reference/lagomorph/extension/extension.cpp:29-102) and its pybind module binds (:175-189), as thin
// forwarders onto the C ABI of liblagomorph_hip.so (include/lagomorph_hip.h).  A maintainer drops the four
// cuda/*.cu translation units from setup.py:25-30, adds this file, and links the extension against
// liblagomorph_hip.so; extension.cpp (argument checks + PYBIND11_MODULE) stays as it is.  `set_debug_mode`
// (the 13th name) lives in extension.cpp and only sets `lagomorph_debug_mode`; every forwarder hands that
// flag to the library before it launches.
//
// Each forwarder allocates its outputs where the reference's host function does (the library overwrites
// them in full, so at::empty replaces at::zeros), passes torch's CURRENT stream, and turns a non-zero
// return code into the c10::Error the reference's TORCH_CHECKs raise.
//
// tests/test_abi_and_host.py compiles this file against the installed torch headers (CPU container, no
// GPU needed): the signatures below are checked against extension.cpp's forward declarations by the
// compiler when both are in one build.
#include <torch/extension.h>

#include <c10/hip/HIPStream.h>

#include <algorithm>
#include <vector>

#include "lagomorph_hip.h"

extern bool lagomorph_debug_mode;  // extension.cpp:26

namespace {

void *current_stream() { return (void *)c10::hip::getCurrentHIPStream().stream(); }

void check(int rc) { TORCH_CHECK(rc == LAGO_OK, lago_last_error()); }

void sync_debug_flag() { lago_set_debug(lagomorph_debug_mode ? 1 : 0); }

int spatial_dim(const at::Tensor &t) { return (int)t.dim() - 2; }
int64_t nz_of(const at::Tensor &t) { return t.dim() == 5 ? t.size(4) : 1; }

// calls F_f32 or F_f64 with the tensor's scalar type (the reference: AT_DISPATCH_FLOATING_TYPES)
#define LAGO_DISPATCH(T, NAME, ...)                                                     \
    do {                                                                                \
        sync_debug_flag();                                                              \
        if ((T).scalar_type() == at::kFloat) {                                          \
            using Real = float;                                                         \
            check(NAME##_f32(__VA_ARGS__));                                             \
        } else if ((T).scalar_type() == at::kDouble) {                                  \
            using Real = double;                                                        \
            check(NAME##_f64(__VA_ARGS__));                                             \
        } else {                                                                        \
            TORCH_CHECK(false, #NAME ": only float32 and float64 are supported");       \
        }                                                                               \
    } while (0)
#define P(t) ((t).defined() && (t).numel() ? (t).data_ptr<Real>() : (Real *)nullptr)

}  // namespace

// cuda/interp.cu:80-130
at::Tensor interp_cuda_forward(at::Tensor Iv, at::Tensor u, double dt) {
    const int d = spatial_dim(Iv);
    TORCH_CHECK(d == 2 || d == 3, "Only two- and three-dimensional interpolation is supported");
    const int64_t nn = std::max(u.size(0), Iv.size(0));
    auto sizes = Iv.sizes().vec();
    sizes[0] = nn;
    auto out = at::empty(sizes, Iv.options());
    LAGO_DISPATCH(Iv, lago_interp_forward, P(out), P(Iv), P(u), dt, d, nn, Iv.size(1), Iv.size(2), Iv.size(3), nz_of(Iv),
                  Iv.size(0) < nn, current_stream());
    return out;
}

// cuda/interp.cu:246-313
std::vector<at::Tensor> interp_cuda_backward(at::Tensor grad_out, at::Tensor Iv, at::Tensor u, double dt, bool need_I,
                                             bool need_u) {
    const int d = spatial_dim(Iv);
    TORCH_CHECK(d == 2 || d == 3, "Only two- and three-dimensional interpolation is supported");
    const int64_t nn = std::max(u.size(0), Iv.size(0));
    auto d_I = at::empty_like(Iv), d_u = at::empty_like(u);
    LAGO_DISPATCH(Iv, lago_interp_backward, P(d_I), P(d_u), P(grad_out), P(Iv), P(u), dt, d, nn, Iv.size(1), Iv.size(2),
                  Iv.size(3), nz_of(Iv), Iv.size(0) < nn, need_I, need_u, current_stream());
    return {d_I, d_u};
}

// cuda/interp.cu:351-381
at::Tensor interp_hessian_diagonal_image(at::Tensor Iv, at::Tensor u, double dt) {
    TORCH_CHECK(Iv.dim() == 4, "interp_hessian_diagonal_image is only implemented for two-dimensional images");
    const int64_t nn = std::max(u.size(0), Iv.size(0));
    auto out = at::empty_like(Iv);
    LAGO_DISPATCH(Iv, lago_interp_hessian_diagonal_image, P(out), P(u), dt, Iv.size(0), nn, Iv.size(1), Iv.size(2),
                  Iv.size(3), current_stream());
    return out;
}

// cuda/affine.cu:114-169
at::Tensor affine_interp_cuda_forward(at::Tensor I, at::Tensor A, at::Tensor T) {
    const int d = spatial_dim(I);
    TORCH_CHECK(A.size(0) == T.size(0), "A and T must have same first dimension");
    const int64_t nn = A.size(0);
    auto sizes = I.sizes().vec();
    sizes[0] = nn;
    auto out = at::empty(sizes, I.options());
    LAGO_DISPATCH(I, lago_affine_interp_forward, P(out), P(I), P(A), P(T), d, nn, I.size(1), I.size(2), I.size(3), nz_of(I),
                  I.size(0) == 1 && nn > 1, current_stream());
    return out;
}

// cuda/affine.cu:538-610 (unneeded gradients are size-0 tensors)
std::vector<at::Tensor> affine_interp_cuda_backward(at::Tensor grad_out, at::Tensor I, at::Tensor A, at::Tensor T,
                                                    bool need_I, bool need_A, bool need_T) {
    const int d = spatial_dim(I);
    TORCH_CHECK(I.size(1) == grad_out.size(1), "I and grad_out must have same number of channels");
    TORCH_CHECK(A.size(0) == T.size(0), "A and T must have same first dimension");
    const int64_t nn = grad_out.size(0);
    auto d_I = need_I ? at::empty_like(I) : at::zeros({0}, I.options());
    auto d_A = need_A ? at::empty_like(A) : at::zeros({0}, I.options());
    auto d_T = need_T ? at::empty_like(T) : at::zeros({0}, I.options());
    LAGO_DISPATCH(I, lago_affine_interp_backward, P(d_I), P(d_A), P(d_T), P(grad_out), P(I), P(A), P(T), d, nn, I.size(1),
                  I.size(2), I.size(3), nz_of(I), I.size(0) == 1 && nn > 1, need_I, need_A, need_T, current_stream());
    return {d_I, d_A, d_T};
}

// cuda/affine.cu:683-734
at::Tensor regrid_forward(at::Tensor I, std::vector<int> shape, std::vector<double> origin, std::vector<double> spacing) {
    const int d = spatial_dim(I);
    TORCH_CHECK(d == 2 || d == 3, "Only two- and three-dimensional regridding is supported");
    TORCH_CHECK((int)shape.size() == d, "Shape should be vector of size d (not 2+d)");
    TORCH_CHECK((int)origin.size() == d, "Origin should be vector of size d (not 2+d)");
    TORCH_CHECK((int)spacing.size() == d, "Spacing should be vector of size d (not 2+d)");
    std::vector<int64_t> sizes = {I.size(0), I.size(1)};
    for (int s : shape) sizes.push_back(s);
    auto out = at::empty(sizes, I.options());
    origin.resize(3, 0.0);
    spacing.resize(3, 0.0);
    LAGO_DISPATCH(I, lago_regrid_forward, P(out), P(I), d, I.size(0), I.size(1), I.size(2), I.size(3), nz_of(I), shape[0],
                  shape[1], d == 3 ? shape[2] : 1, origin.data(), spacing.data(), current_stream());
    return out;
}

// cuda/affine.cu:802-855
at::Tensor regrid_backward(at::Tensor grad_out, std::vector<int> inshape, std::vector<int> shape,
                           std::vector<double> origin, std::vector<double> spacing) {
    const int d = spatial_dim(grad_out);
    TORCH_CHECK(d == 2 || d == 3, "Only two- and three-dimensional regridding is supported");
    TORCH_CHECK((int)inshape.size() == d, "Input shape should be vector of size d (not 2+d)");
    TORCH_CHECK((int)shape.size() == d, "Shape should be vector of size d (not 2+d)");
    std::vector<int64_t> sizes = {grad_out.size(0), grad_out.size(1)};
    for (int s : inshape) sizes.push_back(s);
    auto d_I = at::empty(sizes, grad_out.options());
    origin.resize(3, 0.0);
    spacing.resize(3, 0.0);
    LAGO_DISPATCH(grad_out, lago_regrid_backward, P(d_I), P(grad_out), d, grad_out.size(0), grad_out.size(1), inshape[0],
                  inshape[1], d == 3 ? inshape[2] : 1, shape[0], shape[1], d == 3 ? shape[2] : 1, origin.data(),
                  spacing.data(), current_stream());
    return d_I;
}

// cuda/diff.cu:129-185
at::Tensor jacobian_times_vectorfield_forward(at::Tensor g, at::Tensor v, bool displacement, bool transpose) {
    const int d = spatial_dim(g);
    TORCH_CHECK(g.size(0) == v.size(0), "arguments must have same batch size dimension");
    g = g.contiguous();
    v = v.contiguous();
    auto out = at::empty_like(g);
    LAGO_DISPATCH(g, lago_jtv_forward, P(out), P(g), P(v), displacement, transpose, d, g.size(0), g.size(1), g.size(2),
                  g.size(3), nz_of(g), current_stream());
    return out;
}

// cuda/diff.cu:475-540
std::vector<at::Tensor> jacobian_times_vectorfield_backward(at::Tensor grad_out, at::Tensor v, at::Tensor w,
                                                            bool displacement, bool transpose, bool need_v, bool need_w) {
    const int d = spatial_dim(v);
    TORCH_CHECK(v.size(0) == w.size(0), "arguments must have same batch size dimension");
    grad_out = grad_out.contiguous();
    v = v.contiguous();
    w = w.contiguous();
    auto d_v = at::empty_like(v), d_w = at::empty_like(w);
    LAGO_DISPATCH(v, lago_jtv_backward, P(d_v), P(d_w), P(grad_out), P(v), P(w), displacement, transpose, d, v.size(0),
                  v.size(1), v.size(2), v.size(3), nz_of(v), current_stream());
    return {d_v, d_w};
}

// cuda/diff.cu:634-672
at::Tensor jacobian_times_vectorfield_adjoint_forward(at::Tensor g, at::Tensor v) {
    const int d = spatial_dim(g);
    g = g.contiguous();
    v = v.contiguous();
    auto out = at::empty_like(g);
    LAGO_DISPATCH(g, lago_jtv_adjoint_forward, P(out), P(g), P(v), d, g.size(0), g.size(1), g.size(2), g.size(3), nz_of(g),
                  current_stream());
    return out;
}

// cuda/diff.cu:783-835
std::vector<at::Tensor> jacobian_times_vectorfield_adjoint_backward(at::Tensor grad_out, at::Tensor v, at::Tensor w,
                                                                    bool need_v, bool need_w) {
    const int d = spatial_dim(v);
    grad_out = grad_out.contiguous();
    v = v.contiguous();
    w = w.contiguous();
    auto d_v = at::empty_like(v), d_w = at::empty_like(w);
    LAGO_DISPATCH(v, lago_jtv_adjoint_backward, P(d_v), P(d_w), P(grad_out), P(v), P(w), d, v.size(0), v.size(2), v.size(3),
                  nz_of(v), current_stream());
    return {d_v, d_w};
}

// cuda/metric.cu:308-355 (in place on Fmv: (N, d, nx, ny[, nzc], 2))
void fluid_operator_cuda(at::Tensor Fmv, bool inverse, std::vector<at::Tensor> coslut, std::vector<at::Tensor> sinlut,
                         double alpha, double beta, double gamma) {
    const int d = (int)Fmv.dim() - 3;
    TORCH_CHECK(d == 2 || d == 3, "Only two- and three-dimensional fluid metric is supported");
    for (int i = 0; i < d; ++i) {
        TORCH_CHECK(coslut[i].scalar_type() == Fmv.scalar_type() && sinlut[i].scalar_type() == Fmv.scalar_type(),
                    "Type of LUTs must equal that of image");
        coslut[i] = coslut[i].contiguous();
        sinlut[i] = sinlut[i].contiguous();
    }
    at::Tensor none;
    LAGO_DISPATCH(Fmv, lago_fluid_operator, P(Fmv), inverse, P(coslut[0]), P(sinlut[0]), P(coslut[1]), P(sinlut[1]),
                  P(d == 3 ? coslut[2] : none), P(d == 3 ? sinlut[2] : none), alpha, beta, gamma, d, Fmv.size(0),
                  Fmv.size(2), Fmv.size(3), d == 3 ? Fmv.size(4) : 1, current_stream());
}
