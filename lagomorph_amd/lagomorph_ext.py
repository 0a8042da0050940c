"""``lagomorph_ext`` for MI355X: the reference's 13-function extension surface
(``/root/reference/lagomorph/extension/extension.cpp:175-189``) bound with
ctypes onto ``liblagomorph_hip.so`` (C ABI in ``include/lagomorph_hip.h``).

Same names, argument order, defaults and return conventions as the pybind11
module the reference builds with ``CUDAExtension`` (``setup.py:18-31``):
inputs are borrowed contiguous device tensors, outputs are freshly allocated
tensors, ``fluid_operator`` mutates its first argument, argument violations
raise ``RuntimeError`` (the reference's ``TORCH_CHECK``).  Kernels are launched
on torch's *current* HIP stream (the reference used the legacy default stream).

There is no CPU path: every entry point raises on a CPU tensor, and importing
this module raises if the HIP library has not been built
(``python -m lagomorph_amd.build`` / ``__graft_entry__.build()``).
"""
import ctypes
import math
import threading
import os

import torch  # must be imported first: it loads the process's libamdhip64.so.7

_HERE = os.path.dirname(os.path.abspath(__file__))
# LAGO_HIP_LIBRARY selects another build of the same C ABI (tools/ use the -DLAGO_PROFILING build this way)
LIB_PATH = os.environ.get("LAGO_HIP_LIBRARY") or os.path.join(_HERE, "_lib", "liblagomorph_hip.so")
ABI_VERSION = 5  # LAGO_ABI_VERSION of include/lagomorph_hip.h this binding was written against

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: the HIP extension is not built "
        "(run `python -m lagomorph_amd.build`); lagomorph_amd has no CPU fallback"
    )
_lib = ctypes.CDLL(LIB_PATH)
_lib.lago_abi_version.restype = ctypes.c_int
if _lib.lago_abi_version() != ABI_VERSION:
    # a stale library would be called with shifted arguments (memory corruption, not an error): refuse it
    raise ImportError(
        f"{LIB_PATH} has C-ABI version {_lib.lago_abi_version()}, this binding needs {ABI_VERSION}: "
        "rebuild it with `python -m lagomorph_amd.build -f`"
    )
_lib.lago_last_error.restype = ctypes.c_char_p
_lib.lago_version.restype = ctypes.c_char_p

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_dbl = ctypes.c_double

# argtypes per entry point (both precisions) -- mirrors include/lagomorph_hip.h
_SIGS = {
    "lago_interp_forward": [_vp, _vp, _vp, _dbl, _int, _i64, _i64, _i64, _i64, _i64, _int, _vp],
    "lago_interp_backward": [_vp, _vp, _vp, _vp, _vp, _dbl, _int, _i64, _i64, _i64, _i64, _i64, _int, _int, _int, _vp],
    "lago_interp_hessian_diagonal_image": [_vp, _vp, _dbl, _i64, _i64, _i64, _i64, _i64, _vp],
    "lago_jtv_forward": [_vp, _vp, _vp, _int, _int, _int, _i64, _i64, _i64, _i64, _i64, _vp],
    "lago_jtv_backward": [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _i64, _i64, _i64, _i64, _i64, _vp],
    "lago_jtv_adjoint_forward": [_vp, _vp, _vp, _int, _i64, _i64, _i64, _i64, _i64, _vp],
    "lago_jtv_adjoint_backward": [_vp, _vp, _vp, _vp, _vp, _int, _i64, _i64, _i64, _i64, _vp],
    "lago_fluid_operator": [_vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _dbl, _dbl, _int, _i64, _i64, _i64, _i64, _vp],
    "lago_affine_interp_forward": [_vp, _vp, _vp, _vp, _int, _i64, _i64, _i64, _i64, _i64, _int, _vp],
    "lago_affine_interp_backward": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _i64, _i64, _i64, _i64, _i64, _int, _int,
                                    _int, _int, _vp],
    "lago_regrid_forward": [_vp, _vp, _int, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp],
    "lago_regrid_backward_sep": [_vp, _vp, _vp, _i64, _int, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp],
    "lago_regrid_backward": [_vp, _vp, _int, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp],
    "lago_compose": [_vp, _vp, _vp, _dbl, _dbl, _int, _i64, _i64, _i64, _i64, _vp],
    "lago_lincomb": [_vp, _int, _vp, _vp, _vp, _vp, _dbl, _dbl, _dbl, _dbl, _i64, _vp],
    "lago_Ad_star": [_vp, _vp, _vp, _vp, _int, _i64, _i64, _i64, _i64, _vp],
    "lago_interp_backward_fused": [_vp, _vp, _vp, _vp, _vp, _dbl, _int, _i64, _i64, _i64, _i64, _i64, _int, _int, _int, _int,
                                   _dbl, _vp],
    "lago_jtv_backward_acc": [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _i64, _i64, _i64, _i64, _i64, _int, _vp],
    "lago_ad_star": [_vp, _vp, _vp, _int, _i64, _i64, _i64, _i64, _vp],
    "lago_fluid_metric": [_vp, _vp, _vp, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _dbl, _dbl, _int, _i64, _i64,
                          _i64, _i64, _vp],
    "lago_fluid_metric_scaled": [_vp, _vp, _vp, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _dbl, _dbl, _int, _i64,
                                 _i64, _i64, _i64, _dbl, _vp],
}
_fn = {}
for _name, _args in _SIGS.items():
    for _suf in ("_f32", "_f64"):
        _f = getattr(_lib, _name + _suf, None)
        if _f is None:
            continue
        _f.argtypes = _args
        _f.restype = _int
        _fn[_name + _suf] = _f
_lib.lago_set_debug.argtypes = [_int]


class LagoTuning(ctypes.Structure):
    """include/lagomorph_hip.h: lago_tuning (field order and types as declared there)."""
    _fields_ = [("struct_size", ctypes.c_uint32), ("splat_mode", ctypes.c_int32), ("splat_tile", ctypes.c_int32 * 7),
                ("splat_shear", ctypes.c_int32 * 8), ("splat_shear_mc", ctypes.c_int32), ("splat_mc", ctypes.c_int32),
                ("vector_kernels", ctypes.c_int32), ("launch_order", ctypes.c_int32), ("stencil_tile", ctypes.c_int32),
                ("gather_window", ctypes.c_int32), ("fluid_mode", ctypes.c_int32), ("fluid_xpass_ipw", ctypes.c_int32),
                ("fluid_zy_persist", ctypes.c_int32), ("fluid_xpass_wide", ctypes.c_int32),
                ("fluid_xpass_persist", ctypes.c_int32), ("affine_box", ctypes.c_int32)]


_lib.lago_get_tuning.argtypes = [ctypes.POINTER(LagoTuning)]
_lib.lago_get_tuning.restype = None
_lib.lago_default_tuning.argtypes = [ctypes.POINTER(LagoTuning)]
_lib.lago_default_tuning.restype = None
_lib.lago_set_tuning.argtypes = [ctypes.POINTER(LagoTuning)]
_lib.lago_set_tuning.restype = _int
_tune_lock = threading.Lock()


def _tuning_struct(getter):
    t = LagoTuning()
    t.struct_size = ctypes.sizeof(LagoTuning)
    getter(ctypes.byref(t))
    return t


def _as_dict(t):
    return {n: (list(getattr(t, n)) if hasattr(getattr(t, n), "__len__") else int(getattr(t, n)))
            for n, _ in LagoTuning._fields_ if n != "struct_size"}


def get_tuning():
    """The tuning settings in force (include/lagomorph_hip.h: lago_tuning) as a dict."""
    return _as_dict(_tuning_struct(_lib.lago_get_tuning))


def default_tuning():
    return _as_dict(_tuning_struct(_lib.lago_default_tuning))


def tune(**fields):
    """Change some tuning settings (process-wide, speed only): read the struct, replace the named fields, write it back.
    tune(splat_shear_mc=1), tune(splat_shear=[1, 8, 6, 0, 1, 1, 4, 1024]), ...; tune(**default_tuning()) restores all."""
    with _tune_lock:
        t = _tuning_struct(_lib.lago_get_tuning)
        for k, v in fields.items():
            if k == "struct_size" or k not in dict(LagoTuning._fields_):
                raise KeyError(f"lago_tuning has no field {k!r}")
            cur = getattr(t, k)
            if hasattr(cur, "__len__"):
                v = [int(x) for x in v]
                if len(v) != len(cur):
                    raise ValueError(f"lago_tuning.{k} takes {len(cur)} values")
                for i, x in enumerate(v):
                    cur[i] = x
            else:
                setattr(t, k, int(v))
        if _lib.lago_set_tuning(ctypes.byref(t)) != 0:
            raise RuntimeError(_lib.lago_last_error().decode())


def _suffix(t):
    if t.dtype == torch.float32:
        return "_f32"
    if t.dtype == torch.float64:
        return "_f64"
    raise RuntimeError(f"lagomorph_ext: only float32 and float64 are supported (got {t.dtype})")


def _check_input(x, name):
    # CHECK_INPUT, extension.cpp:8-10
    if not isinstance(x, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not x.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not x.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")


def _same(ref, *others):
    for o in others:
        if o.dtype != ref.dtype:
            raise RuntimeError(f"lagomorph_ext: dtype mismatch ({ref.dtype} vs {o.dtype})")
        if o.device != ref.device:
            raise RuntimeError(f"lagomorph_ext: device mismatch ({ref.device} vs {o.device})")


def _spatial(t):
    d = t.dim() - 2
    if d == 2:
        return 2, t.size(2), t.size(3), 1
    if d == 3:
        return 3, t.size(2), t.size(3), t.size(4)
    return d, 0, 0, 0


def _call(name, ref, *args):
    """Launch on torch's current stream of ref's device; raise RuntimeError on failure."""
    f = _fn[name + _suffix(ref)]
    if ref.device.index != torch.cuda.current_device():
        with torch.cuda.device(ref.device):
            rc = f(*args, torch.cuda.current_stream().cuda_stream)
    else:
        rc = f(*args, torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RuntimeError(_lib.lago_last_error().decode("utf-8", "replace"))


def _ptr(t):
    return t.data_ptr() if t is not None and t.numel() > 0 else None


# ---------------------------------------------------------------------------


def set_debug_mode(mode):
    """extension.cpp:105-107.  In debug mode every launch is followed by a
    stream synchronise and kernel faults raise (the reference printed them)."""
    _lib.lago_set_debug(1 if mode else 0)


def set_splat_mode(mode):
    """0 = global atomics only, 1 = LDS-privatised splat (default)."""
    tune(splat_mode=mode)


def set_splat_tile(tx, ty, tz, mx, my, mz, nthreads):
    tune(splat_tile=[tx, ty, tz, mx, my, mz, nthreads])


def set_splat_shear(on=1, tx=8, ty=6, tz=0, mx=1, my=1, mz=4, nthreads=1024):
    """Sheared-window float32 splat (csrc/splat.hip: splat_shear_kernel): on/off and its tile.  Speed only."""
    tune(splat_shear=[on, tx, ty, tz, mx, my, mz, nthreads])


def set_splat_mc(on):
    """General tiled splat: 1 (default) the multi-channel single-pass form where it applies.  Speed only."""
    tune(splat_mc=1 if on else 0)


def set_splat_shear_mc(mode):
    """Sheared-window splat, several channels with d_u: 2 (default) geometry and d_u sums in registers over the
    channels, 1 the d_u sums only, 0 neither.  Speed only (same d_u bits)."""
    tune(splat_shear_mc=mode)


def set_fluid_tuning(xpass_ipw=0, zy_persist=1, xpass_wide=1, xpass_persist=1):
    """FFT-pass fluid metric: batch items per x-pass workgroup (0 = by launch size), persistent zy kernels for
    planes above 80 KB, 512-thread x pass for the 256-point tile, persistent prefetching x-pass grid.  Speed only."""
    tune(fluid_xpass_persist=xpass_persist, fluid_xpass_ipw=xpass_ipw, fluid_zy_persist=1 if zy_persist else 0,
         fluid_xpass_wide=1 if xpass_wide else 0)


def set_launch_order(alternate):
    """1 (default): successive launches walk their workgroups in alternating directions (a consumer starts on what
    its producer wrote last: Infinity-Cache reuse); 0: always ascending.  Speed only."""
    tune(launch_order=1 if alternate else 0)


def set_stencil_tile(on):
    """1 (default): LDS row-tile stencil kernels (Ad_star, jacobian_times_vectorfield_backward) where shapes allow;
    0: the direct one-lane-per-voxel kernels; 3: row tiles without the compile-time-geometry instantiations for 128^3 /
    160^3 volumes.  Same bits."""
    tune(stencil_tile=on)


_lib.lago_path_launches.restype = ctypes.c_longlong
_lib.lago_path_launches.argtypes = [_int]
PATHS = ("gather_window", "stencil_tile", "vector_gather", "splat_shear", "splat_shear_mc", "splat_tiled", "splat_global",
         "fluid_lds", "fluid_2d", "fluid_xpass", "fluid_rocfft", "splat_2d", "splat_affine_box", "fluid_generic")  # LAGO_PATH_* of include/lagomorph_hip.h, in order


_lib.lago_reversed_launches.restype = ctypes.c_longlong
_lib.lago_reversed_launches.argtypes = []


def reversed_launches():
    """Launches so far that walked their workgroups in descending order (include/lagomorph_hip.h)."""
    return int(_lib.lago_reversed_launches())


def path_launches(name=None):
    """Launches so far per implementation path (a dict), or of one path by name."""
    if name is not None:
        return int(_lib.lago_path_launches(PATHS.index(name)))
    return {p: int(_lib.lago_path_launches(i)) for i, p in enumerate(PATHS)}


def set_gather_window(on):
    """1 (default): float32 3D gathers through an LDS window (compose) where shapes allow; 0: pair gathers
    through the vector L1 only.  Same bits."""
    tune(gather_window=1 if on else 0)


if os.environ.get("LAGO_GATHER_WINDOW") is not None:  # profiling convenience: A/B under rocprofv3 without code changes
    # (parsed defensively: an empty or odd value of a profiling variable must not break `import lagomorph_amd`)
    set_gather_window(os.environ["LAGO_GATHER_WINDOW"].strip().lower() not in ("", "0", "off", "false", "no"))


def set_vector_kernels(on):
    """1 (default): slab-unrolled 3D gather kernels (two voxels per lane) where shapes allow; 0: one-voxel-per-lane kernels only."""
    tune(vector_kernels=1 if on else 0)


def set_fluid_mode(mode):
    """fluid_metric implementation: 3 (default) the tuned LDS-tiled FFT passes where the shape allows and the generic
    hand-written passes for everything else (no rocFFT); 2 the tuned passes with rocFFT fallbacks, 1 rocFFT 2D plan +
    fused x pass, 0 rocFFT 3D plan + operator kernel; 4 = 3 with the generic passes' x transforms and operator as three
    launches instead of the fused one (same bits: the comparison switch of that fusion)."""
    tune(fluid_mode=mode)


def version():
    return _lib.lago_version().decode()


def interp_forward(Iv, u, dt=1.0):
    """extension.cpp:135-143 -> cuda/interp.cu:80-130"""
    _check_input(Iv, "Iv")
    _check_input(u, "u")
    _same(Iv, u)
    dim, nx, ny, nz = _spatial(Iv)
    if dim not in (2, 3):
        raise RuntimeError("Only two- and three-dimensional interpolation is supported")
    if u.dim() != Iv.dim() or u.size(1) != dim or tuple(u.shape[2:]) != tuple(Iv.shape[2:]):
        raise RuntimeError(f"interp_forward: displacement of shape {tuple(u.shape)} does not match image {tuple(Iv.shape)}")
    nn = max(u.size(0), Iv.size(0))
    bc = Iv.size(0) < nn
    if (bc and Iv.size(0) != 1) or u.size(0) != nn:
        raise RuntimeError("interp_forward: batch sizes of I and u are incompatible")
    out = torch.empty((nn, Iv.size(1)) + tuple(Iv.shape[2:]), dtype=Iv.dtype, device=Iv.device)
    _call("lago_interp_forward", Iv, _ptr(out), _ptr(Iv), _ptr(u), float(dt), dim, nn, Iv.size(1), nx, ny, nz, int(bc))
    return out


def interp_backward(grad_out, I, u, dt, need_I, need_u):
    """extension.cpp:145-156 -> cuda/interp.cu:246-313.  Returns [d_I, d_u]; both always allocated."""
    _check_input(grad_out, "grad_out")
    _check_input(I, "I")
    _check_input(u, "u")
    _same(I, u, grad_out)
    dim, nx, ny, nz = _spatial(I)
    if dim not in (2, 3):
        raise RuntimeError("Only two- and three-dimensional interpolation is supported")
    nn = max(u.size(0), I.size(0))
    bc = I.size(0) < nn
    if (bc and I.size(0) != 1) or u.size(0) != nn:
        raise RuntimeError("interp_backward: batch sizes of I and u are incompatible")
    if (tuple(grad_out.shape) != (nn, I.size(1)) + tuple(I.shape[2:]) or u.dim() != I.dim() or u.size(1) != dim
            or tuple(u.shape[2:]) != tuple(I.shape[2:])):
        raise RuntimeError("interp_backward: grad_out / I / u shapes are inconsistent")
    d_I = torch.empty_like(I)
    d_u = torch.empty_like(u)
    _call("lago_interp_backward", I, _ptr(d_I), _ptr(d_u), _ptr(grad_out), _ptr(I), _ptr(u), float(dt), dim, nn,
          I.size(1), nx, ny, nz, int(bc), int(bool(need_I)), int(bool(need_u)))
    return [d_I, d_u]


def interp_hessian_diagonal_image(Iv, u, dt):
    """cuda/interp.cu:351-381 (2D only; accumulates into plane 0 like the reference)."""
    _check_input(Iv, "Iv")
    _check_input(u, "u")
    _same(Iv, u)
    if Iv.dim() != 4 or u.dim() != 4 or u.size(1) != 2:
        raise RuntimeError("interp_hessian_diagonal_image is only implemented for two-dimensional images")
    nn = max(u.size(0), Iv.size(0))
    out = torch.empty_like(Iv)
    _call("lago_interp_hessian_diagonal_image", Iv, _ptr(out), _ptr(u), float(dt), Iv.size(0), nn, Iv.size(1),
          Iv.size(2), Iv.size(3))
    return out


def _check_jtv(g, v):
    for x, nm in ((g, "g"), (v, "v")):
        if not isinstance(x, torch.Tensor) or not x.is_cuda:
            raise RuntimeError(f"{nm} must be a CUDA tensor")
    _same(g, v)


def jacobian_times_vectorfield_forward(g, v, displacement, transpose):
    """cuda/diff.cu:129-185: (Dg + [displacement] I) v, or its transpose."""
    _check_jtv(g, v)
    g, v = g.contiguous(), v.contiguous()
    dim, nx, ny, nz = _spatial(g)
    if dim not in (2, 3):
        raise RuntimeError("Only two- and three-dimensional jacobian times vectorfield is supported")
    if g.size(0) != v.size(0):
        raise RuntimeError("arguments must have same batch size dimension")
    if v.size(1) != dim or tuple(v.shape[2:]) != tuple(g.shape[2:]):
        raise RuntimeError("vector field is of wrong dimension")
    out = torch.empty_like(g)
    _call("lago_jtv_forward", g, _ptr(out), _ptr(g), _ptr(v), int(bool(displacement)), int(bool(transpose)), dim,
          g.size(0), g.size(1), nx, ny, nz)
    return out


def jacobian_times_vectorfield_backward(grad_out, v, w, displacement, transpose, need_v, need_w, d_v=None):
    """cuda/diff.cu:475-540 (need_v / need_w are ignored there: both gradients are always computed).  Beyond the
    reference: with `d_v` given, the gradient is added onto it in place (lago_jtv_backward_acc)."""
    _check_jtv(v, w)
    _check_jtv(v, grad_out)
    grad_out, v, w = grad_out.contiguous(), v.contiguous(), w.contiguous()
    dim, nx, ny, nz = _spatial(v)
    if dim not in (2, 3):
        raise RuntimeError("Only two- and three-dimensional jacobian times vectorfield is supported")
    if v.size(0) != w.size(0):
        raise RuntimeError("arguments must have same batch size dimension")
    if w.size(1) != dim or tuple(w.shape[2:]) != tuple(v.shape[2:]) or grad_out.shape != v.shape:
        raise RuntimeError("vector field is of wrong dimension")
    d_w = torch.empty_like(w)
    if d_v is not None:
        _check_input(d_v, "d_v")
        _same(v, d_v)
        if d_v.shape != v.shape:
            raise RuntimeError("jacobian_times_vectorfield_backward: d_v must have the shape of v")
        _call("lago_jtv_backward_acc", v, _ptr(d_v), _ptr(d_w), _ptr(grad_out), _ptr(v), _ptr(w), int(bool(displacement)),
              int(bool(transpose)), dim, v.size(0), v.size(1), nx, ny, nz, 1)
        return [d_v, d_w]
    d_v = torch.empty_like(v)
    _call("lago_jtv_backward", v, _ptr(d_v), _ptr(d_w), _ptr(grad_out), _ptr(v), _ptr(w), int(bool(displacement)),
          int(bool(transpose)), dim, v.size(0), v.size(1), nx, ny, nz)
    return [d_v, d_w]


def jacobian_times_vectorfield_adjoint_forward(g, v):
    """cuda/diff.cu:634-672"""
    _check_jtv(g, v)
    g, v = g.contiguous(), v.contiguous()
    dim, nx, ny, nz = _spatial(g)
    if dim not in (2, 3):
        raise RuntimeError("Only two- and three-dimensional jacobian times vectorfield is supported")
    if v.size(1) != dim or v.size(0) != g.size(0) or tuple(v.shape[2:]) != tuple(g.shape[2:]):
        raise RuntimeError("vector field is of wrong dimension")
    out = torch.empty_like(g)
    _call("lago_jtv_adjoint_forward", g, _ptr(out), _ptr(g), _ptr(v), dim, g.size(0), g.size(1), nx, ny, nz)
    return out


def jacobian_times_vectorfield_adjoint_backward(grad_out, v, w, need_v, need_w):
    """cuda/diff.cu:783-835"""
    _check_jtv(v, w)
    _check_jtv(v, grad_out)
    grad_out, v, w = grad_out.contiguous(), v.contiguous(), w.contiguous()
    dim, nx, ny, nz = _spatial(v)
    if dim not in (2, 3):
        raise RuntimeError("Only two- and three-dimensional jacobian times vectorfield is supported")
    if w.size(1) != dim or v.size(1) != dim or v.shape != w.shape or grad_out.shape != v.shape:
        raise RuntimeError("vector field is of wrong dimension")
    d_v, d_w = torch.empty_like(v), torch.empty_like(w)
    _call("lago_jtv_adjoint_backward", v, _ptr(d_v), _ptr(d_w), _ptr(grad_out), _ptr(v), _ptr(w), dim, v.size(0), nx,
          ny, nz)
    return [d_v, d_w]


def fluid_operator(Fmv, inverse, cosluts, sinluts, alpha, beta, gamma):
    """extension.cpp:158-173 -> cuda/metric.cu:308-355.  In place on Fmv (N, d, nx, ny[, nzc], 2)."""
    _check_input(Fmv, "Fmv")
    dim = Fmv.dim() - 3
    if len(cosluts) != dim:
        raise RuntimeError(f"Must provide same number cosine LUTs ({len(cosluts)}) as spatial dimension '{dim}'")
    if len(sinluts) != dim:
        raise RuntimeError(f"Must provide same number sine LUTs ({len(sinluts)}) as spatial dimension '{dim}'")
    if dim not in (2, 3):
        raise RuntimeError("Only two- and three-dimensional fluid metric is supported")
    if Fmv.size(1) != dim or Fmv.size(-1) != 2:
        raise RuntimeError("Vector field has incorrect shape for dimension")
    luts = []
    for d in range(dim):
        for t in (cosluts[d], sinluts[d]):
            if t.dtype != Fmv.dtype:
                raise RuntimeError("Type of LUTs must equal that of image")
            if not t.is_cuda or t.device != Fmv.device or t.numel() != Fmv.size(2 + d):
                raise RuntimeError("fluid_operator: LUT on wrong device or of wrong length")
            luts.append(t.contiguous())
    sh = [Fmv.size(2 + d) for d in range(dim)] + [1] * (3 - dim)
    p = [_ptr(t) for t in luts] + [None] * (6 - 2 * dim)
    _call("lago_fluid_operator", Fmv, _ptr(Fmv), int(bool(inverse)), *p, float(alpha), float(beta), float(gamma), dim,
          Fmv.size(0), sh[0], sh[1], sh[2])
    return None


def fluid_cache_clear():
    """Drop the library's cached per-frequency coefficient tables."""
    _lib.lago_fluid_cache_clear()


def fluid_cache_entries():
    return int(_lib.lago_fluid_cache_entries())


def fft_plan_state():
    """(cached rocFFT fallback plans, how many of them are currently verified by the spot check)."""
    a, b = ctypes.c_int(0), ctypes.c_int(0)
    _lib.lago_fft_plan_state(ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


def fluid_metric(mv, inverse, cosluts, sinluts, alpha, beta, gamma, lut_generation=0, out_scale=1.0):
    """Whole FluidMetricOperator.forward (metric.py:11-19) in one call: irfft(L^(+-2) rfft(mv)).
    Not part of the reference's extension surface.  Returns a new tensor; mv is not modified.
    `lut_generation` names the CONTENTS of the LUTs (see include/lagomorph_hip.h): non-zero lets the
    library cache its coefficient table under that number; 0 takes the table-free path.
    `out_scale`: a factor on the result, applied to the finished value (the bits of `result * out_scale`) inside the
    last kernel where that kernel can take it (lago_fluid_metric_scaled)."""
    _check_input(mv, "mv")
    dim, nx, ny, nz = _spatial(mv)
    if dim not in (2, 3):
        raise RuntimeError("Only two- and three-dimensional fluid metric is supported")
    if mv.size(1) != dim:
        raise RuntimeError("Vector field has incorrect shape for dimension")
    if len(cosluts) != dim or len(sinluts) != dim:
        raise RuntimeError(f"Must provide same number of LUTs as spatial dimension '{dim}'")
    csh = list(mv.shape[2:])
    csh[-1] = csh[-1] // 2 + 1
    luts = []
    for d in range(dim):
        for t in (cosluts[d], sinluts[d]):
            if t.dtype != mv.dtype:
                raise RuntimeError("Type of LUTs must equal that of image")
            if not t.is_cuda or t.device != mv.device or t.numel() != csh[d]:
                raise RuntimeError("fluid_metric: LUT on wrong device or of wrong length")
            luts.append(t.contiguous())
    out = torch.empty_like(mv)
    work = torch.empty((mv.size(0), dim, *csh, 2), dtype=mv.dtype, device=mv.device)
    p = [_ptr(t) for t in luts] + [None] * (6 - 2 * dim)
    if float(out_scale) != 1.0:
        _call("lago_fluid_metric_scaled", mv, _ptr(out), _ptr(mv), _ptr(work), int(lut_generation), int(bool(inverse)), *p,
              float(alpha), float(beta), float(gamma), dim, mv.size(0), nx, ny, nz, float(out_scale))
        return out
    _call("lago_fluid_metric", mv, _ptr(out), _ptr(mv), _ptr(work), int(lut_generation), int(bool(inverse)), *p, float(alpha), float(beta),
          float(gamma), dim, mv.size(0), nx, ny, nz)
    return out


def affine_interp_forward(I, A, T):
    """extension.cpp:109-118 -> cuda/affine.cu:114-169.  The reference falls back to
    cpu/affine.cpp for CPU tensors; this build is HIP-only and raises instead."""
    _check_input(I, "I")
    _check_input(A, "A")
    _check_input(T, "T")
    _same(I, A, T)
    if A.size(0) != T.size(0):
        raise RuntimeError("A and T must have same first dimension")
    dim, nx, ny, nz = _spatial(I)
    if dim not in (2, 3):
        raise RuntimeError("Only two- and three-dimensional affine interpolation is supported")
    if tuple(A.shape[1:]) != (dim, dim) or tuple(T.shape[1:]) != (dim,):
        raise RuntimeError("affine_interp_forward: A must be (N, d, d) and T (N, d)")
    nn = A.size(0)
    bc = I.size(0) == 1 and nn > 1
    if not bc and I.size(0) != nn:
        raise RuntimeError("affine_interp_forward: batch sizes of I and A are incompatible")
    out = torch.empty((nn, I.size(1)) + tuple(I.shape[2:]), dtype=I.dtype, device=I.device)
    _call("lago_affine_interp_forward", I, _ptr(out), _ptr(I), _ptr(A), _ptr(T), dim, nn, I.size(1), nx, ny, nz, int(bc))
    return out


def affine_interp_backward(grad_out, I, A, T, need_I, need_A, need_T):
    """extension.cpp:120-133 -> cuda/affine.cu:538-610.  Unneeded gradients are size-0 tensors."""
    _check_input(grad_out, "grad_out")
    _check_input(I, "I")
    _check_input(A, "A")
    _check_input(T, "T")
    _same(I, A, T, grad_out)
    if I.size(1) != grad_out.size(1):
        raise RuntimeError("I and grad_out must have same number of channels")
    if A.size(0) != T.size(0):
        raise RuntimeError("A and T must have same first dimension")
    dim, nx, ny, nz = _spatial(I)
    if dim not in (2, 3):
        raise RuntimeError("Only two- and three-dimensional affine interpolation is supported")
    nn = grad_out.size(0)
    bc = I.size(0) == 1 and nn > 1
    if A.size(0) != nn or tuple(grad_out.shape[2:]) != tuple(I.shape[2:]) or (not bc and I.size(0) != nn):
        raise RuntimeError("affine_interp_backward: grad_out / I / A shapes are inconsistent")
    empty = lambda: torch.zeros((0,), dtype=I.dtype, device=I.device)
    d_I = torch.empty_like(I) if need_I else empty()
    d_A = torch.empty_like(A) if need_A else empty()
    d_T = torch.empty_like(T) if need_T else empty()
    _call("lago_affine_interp_backward", I, _ptr(d_I), _ptr(d_A), _ptr(d_T), _ptr(grad_out), _ptr(I), _ptr(A), _ptr(T),
          dim, nn, I.size(1), nx, ny, nz, int(bc), int(bool(need_I)), int(bool(need_A)), int(bool(need_T)))
    return [d_I, d_A, d_T]


def _overlaps(a, b):
    """Do the byte ranges of two contiguous tensors intersect?  (a partially overlapping view counts: the kernel
    gathers from its inputs while it writes `out`)"""
    a0, b0 = a.data_ptr(), b.data_ptr()
    return a.numel() > 0 and b.numel() > 0 and a0 < b0 + b.numel() * b.element_size() and b0 < a0 + a.numel() * a.element_size()


def compose(u, v, ds=1.0, dt=1.0, out=None):
    """Fused deform.compose (deform.py:53-55): ds*u + dt*interp(v, u, dt=ds) in one kernel.
    Not part of the reference's extension surface; u and v are (N, d, *spatial) vector fields.  `out`: a contiguous
    tensor of their shape to write into (it must not alias u or v)."""
    _check_input(u, "u")
    _check_input(v, "v")
    _same(u, v)
    dim, nx, ny, nz = _spatial(u)
    if dim not in (2, 3):
        raise RuntimeError("Only two- and three-dimensional interpolation is supported")
    if u.shape != v.shape or u.size(1) != dim:
        raise RuntimeError("compose: u and v must be vector fields of the same shape")
    if out is None:
        out = torch.empty_like(u)
    else:
        _check_input(out, "out")
        _same(u, out)
        if out.shape != u.shape or _overlaps(out, u) or _overlaps(out, v):
            raise RuntimeError("compose: out must have the shape of u and must not alias an input")
    _call("lago_compose", u, _ptr(out), _ptr(u), _ptr(v), float(ds), float(dt), dim, u.size(0), nx, ny, nz)
    return out


def lincomb(terms, out=None):
    """out = c0*x0 + c1*x1 + ... for 1 to 4 (coefficient, tensor) pairs of one shape and dtype, in one pass
    (left to right, one fma per term).  `out` may be one of the inputs (in place); a new tensor otherwise.
    Not part of the reference's extension surface: the elementwise sums of lddmm_step (lddmm.py:300-325)."""
    if not 1 <= len(terms) <= 4:
        raise RuntimeError("lincomb: 1 to 4 terms")
    xs = [x for _, x in terms]
    for x in xs:
        _check_input(x, "x")
        _same(xs[0], x)
        if x.shape != xs[0].shape:
            raise RuntimeError("lincomb: shapes differ")
    if out is None:
        out = torch.empty_like(xs[0])
    else:
        _check_input(out, "out")
        _same(xs[0], out)
        if out.shape != xs[0].shape:
            raise RuntimeError("lincomb: shapes differ")
    ptrs = [_ptr(x) for x in xs] + [None] * (4 - len(xs))
    cs = [float(c) for c, _ in terms] + [0.0] * (4 - len(terms))
    _call("lago_lincomb", xs[0], _ptr(out), len(terms), *ptrs, *cs, xs[0].numel())
    return out


def interp_backward_fused(grad_out, I, u, dt, need_I, d_u=None, addgo=None, d_I=None):
    """interp_backward whose d_u sum starts from the contents of `d_u` (given: accumulated in place and returned)
    or from addgo * grad_out (needs as many channels as dimensions) instead of zero -- the chain-rule additions of
    compose's and Ad_star's backward without their extra passes; with `d_I` given the splat is added onto it
    instead of onto zeros.  Returns [d_I, d_u].  Not part of the reference's extension surface
    (include/lagomorph_hip.h: lago_interp_backward_fused)."""
    _check_input(grad_out, "grad_out")
    _check_input(I, "I")
    _check_input(u, "u")
    _same(I, u, grad_out)
    dim, nx, ny, nz = _spatial(I)
    if dim not in (2, 3):
        raise RuntimeError("Only two- and three-dimensional interpolation is supported")
    nn = max(u.size(0), I.size(0))
    bc = I.size(0) < nn
    if (bc and I.size(0) != 1) or u.size(0) != nn:
        raise RuntimeError("interp_backward: batch sizes of I and u are incompatible")
    if (tuple(grad_out.shape) != (nn, I.size(1)) + tuple(I.shape[2:]) or u.dim() != I.dim() or u.size(1) != dim
            or tuple(u.shape[2:]) != tuple(I.shape[2:])):
        raise RuntimeError("interp_backward: grad_out / I / u shapes are inconsistent")
    if (d_u is None) == (addgo is None):
        raise RuntimeError("interp_backward_fused: give exactly one of d_u (accumulate) and addgo")
    if d_u is not None:
        _check_input(d_u, "d_u")
        _same(u, d_u)
        if d_u.shape != u.shape:
            raise RuntimeError("interp_backward_fused: d_u must have the shape of u")
        mode, ag = 1, 0.0
    else:
        if I.size(1) != dim:
            raise RuntimeError("interp_backward_fused: addgo needs as many channels as dimensions")
        d_u = torch.empty_like(u)
        mode, ag = 2, float(addgo)
    imode = 0
    if d_I is not None:
        _check_input(d_I, "d_I")
        _same(I, d_I)
        if d_I.shape != I.shape or not need_I:
            raise RuntimeError("interp_backward_fused: d_I must have the shape of I (and need_I be set)")
        imode = 1
    else:
        d_I = torch.empty_like(I)
    _call("lago_interp_backward_fused", I, _ptr(d_I), _ptr(d_u), _ptr(grad_out), _ptr(I), _ptr(u), float(dt), dim, nn,
          I.size(1), nx, ny, nz, int(bc), int(bool(need_I)), imode, mode, ag)
    return [d_I, d_u]


def Ad_star(phiinv, m, save_resampled=False):
    """Fused adjrep.Ad_star (adjrep.py:86-97): jacobian_times_vectorfield(phiinv, interp(m, phiinv),
    displacement=True) in one kernel.  Not part of the reference's extension surface.  With save_resampled the
    resampled momentum interp(m, phiinv) is returned as well: (out, mphiinv)."""
    _check_input(phiinv, "phiinv")
    _check_input(m, "m")
    _same(phiinv, m)
    dim, nx, ny, nz = _spatial(phiinv)
    if dim not in (2, 3):
        raise RuntimeError("Only two- and three-dimensional fields are supported")
    if phiinv.shape != m.shape or m.size(1) != dim:
        raise RuntimeError("Ad_star: phiinv and m must be vector fields of the same shape")
    out = torch.empty_like(m)
    mphi = torch.empty_like(m) if save_resampled else None
    _call("lago_Ad_star", m, _ptr(out), _ptr(mphi), _ptr(phiinv), _ptr(m), dim, m.size(0), nx, ny, nz)
    return (out, mphi) if save_resampled else out


def ad_star(v, m):
    """Fused adjrep.ad_star (adjrep.py:69-83): jacobian_times_vectorfield(v, m, transpose=True) minus
    jacobian_times_vectorfield_adjoint(m, v) in one kernel.  Not part of the reference's extension surface."""
    _check_input(v, "v")
    _check_input(m, "m")
    _same(v, m)
    dim, nx, ny, nz = _spatial(v)
    if dim not in (2, 3):
        raise RuntimeError("Only two- and three-dimensional jacobian times vectorfield is supported")
    if v.shape != m.shape or m.size(1) != dim:
        raise RuntimeError("ad_star: v and m must be vector fields of the same shape")
    out = torch.empty_like(m)
    _call("lago_ad_star", m, _ptr(out), _ptr(v), _ptr(m), dim, m.size(0), nx, ny, nz)
    return out


def _vec3(x, dim, what):
    x = list(x)
    if len(x) != dim:
        raise RuntimeError(f"{what} should be vector of size d (not 2+d)")
    return (ctypes.c_double * 3)(*([float(v) for v in x] + [0.0] * (3 - dim)))


def regrid_forward(I, shape, origin, spacing):
    """cuda/affine.cu:683-734"""
    _check_input(I, "I")
    dim, nx, ny, nz = _spatial(I)
    if dim not in (2, 3):
        raise RuntimeError("Only two- and three-dimensional regridding is supported")
    shape = [int(s) for s in shape]
    if len(shape) != dim:
        raise RuntimeError("Shape should be vector of size d (not 2+d)")
    O, S = _vec3(origin, dim, "Origin"), _vec3(spacing, dim, "Spacing")
    N = shape + [1] * (3 - dim)
    out = torch.empty(tuple(I.shape[:2]) + tuple(shape), dtype=I.dtype, device=I.device)
    _call("lago_regrid_forward", I, _ptr(out), _ptr(I), dim, I.size(0), I.size(1), nx, ny, nz, N[0], N[1], N[2], O, S)
    return out


# 1 (default): regrid_backward runs axis by axis in gather form (lago_regrid_backward_sep: no atomics, no memset, results
# independent of the launch); 0: the reference's splat (LDS-privatised / global atomics by lago_tuning.splat_mode)
REGRID_BACKWARD_SEPARABLE = 1


def regrid_backward(grad_out, inshape, shape, origin, spacing):
    """cuda/affine.cu:802-855"""
    _check_input(grad_out, "grad_out")
    dim = grad_out.dim() - 2
    if dim not in (2, 3):
        raise RuntimeError("Only two- and three-dimensional regridding is supported")
    inshape = [int(s) for s in inshape]
    shape = [int(s) for s in shape]
    if len(inshape) != dim:
        raise RuntimeError("Input shape should be vector of size d (not 2+d)")
    if len(shape) != dim or list(grad_out.shape[2:]) != shape:
        raise RuntimeError("Shape should be vector of size d (not 2+d)")
    O, S = _vec3(origin, dim, "Origin"), _vec3(spacing, dim, "Spacing")
    n3 = inshape + [1] * (3 - dim)
    N = shape + [1] * (3 - dim)
    d_I = torch.empty(tuple(grad_out.shape[:2]) + tuple(inshape), dtype=grad_out.dtype, device=grad_out.device)
    if REGRID_BACKWARD_SEPARABLE and all(S[d] > 0 for d in range(dim)) and d_I.numel() and grad_out.numel():
        # axis by axis in gather form (no atomics, deterministic): two temporaries, each as large as the biggest
        # intermediate (the axes go from grad_out's extents to d_I's one at a time, slowest axis first)
        planes = grad_out.size(0) * grad_out.size(1)
        cur, half = list(shape), 0
        for a in range(dim - 1):
            cur[a] = inshape[a]
            half = max(half, planes * math.prod(cur))
        ws = torch.empty((2 * half,), dtype=grad_out.dtype, device=grad_out.device) if half else None
        _call("lago_regrid_backward_sep", grad_out, _ptr(d_I), _ptr(grad_out), _ptr(ws), 2 * half, dim, grad_out.size(0),
              grad_out.size(1), n3[0], n3[1], n3[2], N[0], N[1], N[2], O, S)
        return d_I
    _call("lago_regrid_backward", grad_out, _ptr(d_I), _ptr(grad_out), dim, grad_out.size(0), grad_out.size(1), n3[0],
          n3[1], n3[2], N[0], N[1], N[2], O, S)
    return d_I
