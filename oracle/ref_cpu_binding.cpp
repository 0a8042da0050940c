// TEST INFRASTRUCTURE ONLY.
// Compiles the reference's own CPU path, /root/reference/lagomorph/extension/cpu/affine.cpp,
// from where it lies (no reference source is copied into this repo) and binds
// its one entry point, affine_interp_cpu_forward (cpu/affine.cpp:129-169).
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

bool lagomorph_debug_mode = false;  // extension.cpp:26 (declared extern in include/defs.h:15)

#include LAGOMORPH_REF_CPU_AFFINE

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("affine_interp_cpu_forward", &affine_interp_cpu_forward,
          "reference cpu/affine.cpp:129 affine_interp_cpu_forward");
}
