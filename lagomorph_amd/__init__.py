"""lagomorph_amd -- MI355X-native LDDMM hot path behind lagomorph's operator surface.

Importing this package requires the built HIP library (no CPU fallback).
"""
import sys as _sys

# `python -m lagomorph_amd.build` imports this package before it runs the builder; with no library yet
# that must not fail (everything else does: there is no CPU fallback).
_BUILDING = "lagomorph_amd.build" in getattr(_sys, "orig_argv", [])

if not _BUILDING:
    from .adjrep import *  # noqa: F401,F403
    from .adjrep import Ad, Ad_dagger, Ad_star, ad, ad_dagger, ad_star, sym, sym_dagger  # noqa: F401
    from .affine import (AffineInterp, AffineInterpFunction, RegridFunction, RegridModule, StandardizedDataset,  # noqa: F401
                         affine_atlas, affine_interp, batch_average, load_affine_atlas, save_affine_atlas,
                         affine_inverse, det_2x2, invert_2x2, invert_3x3, regrid, rigid_inverse, rotation_exp_map)
    from .deform import (InterpFunction, compose, compose_disp_vel, compose_vel_disp, identity, interp,  # noqa: F401
                         interp_hessian_diagonal_image)
    from .diff import (JacobianTimesVectorFieldAdjointFunction, JacobianTimesVectorFieldFunction,  # noqa: F401
                       jacobian_times_vectorfield, jacobian_times_vectorfield_adjoint)
    from .lagomorph_ext import set_debug_mode  # noqa: F401
    from .lddmm import EPDiff_step, LDDMMAtlasBuilder, expmap, expmap_advect, lddmm_step, shard_indices  # noqa: F401
    from .metric import FluidMetric, FluidMetricOperator, Metric  # noqa: F401

    __version__ = "0.1.0"
