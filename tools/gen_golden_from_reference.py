#!/usr/bin/env python3
"""Generate tests/golden/ref_python.npz by running the REFERENCE's own Python layers
(/root/reference/lagomorph/{deform,diff,metric,adjrep,lddmm,affine}.py, imported from where they
lie -- nothing is copied) on seeded inputs.

The reference's compiled extension `lagomorph_ext` is CUDA-only and cannot be built in this image,
so the CPU oracle (oracle/lago_oracle.py: OracleExt) stands in for it; the oracle is pinned to the
reference kernels separately (tests/test_oracle_kat.py, tests/test_oracle_ref.py).  What these
fixtures pin is therefore everything ABOVE the extension boundary: the autograd.Function plumbing
(what is saved, which gradients are returned), FluidMetric's LUT construction and FFT convention,
the adjoint-representation formulas, compose*, EPDiff_step / expmap / expmap_advect, regrid's
displacement scaling -- i.e. rows a11-a13 of SURVEY.md section 8.

`torch.rfft` / `torch.irfft` (removed in torch 1.8, used by metric.py:17-19,25-33) are provided as
thin aliases of torch.fft.rfftn/irfftn(norm="ortho"); `h5py` (absent here, imported by lddmm.py via
data.py) is stubbed with an empty module.  Run in the build container only:

    python tools/gen_golden_from_reference.py
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("LAGOMORPH_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

from oracle.lago_oracle import OracleExt  # noqa: E402


def import_reference():
    sys.modules["lagomorph_ext"] = OracleExt()
    sys.modules.setdefault("h5py", types.ModuleType("h5py"))

    def rfft(x, nd, normalized=False, onesided=True):
        assert normalized and onesided
        return torch.view_as_real(torch.fft.rfftn(x, dim=tuple(range(-nd, 0)), norm="ortho")).contiguous()

    def irfft(X, nd, normalized=False, onesided=True, signal_sizes=None):
        assert normalized and onesided
        return torch.fft.irfftn(torch.view_as_complex(X), s=tuple(signal_sizes), dim=tuple(range(-nd, 0)), norm="ortho")

    torch.rfft, torch.irfft = rfft, irfft
    pkg = types.ModuleType("lagomorph")
    pkg.__path__ = [os.path.join(REF, "lagomorph")]
    sys.modules["lagomorph"] = pkg
    import importlib

    mods = {}
    for name in ("utils", "deform", "diff", "metric", "adjrep", "affine", "lddmm"):
        mods[name] = importlib.import_module("lagomorph." + name)
    return mods


def main():
    m = import_reference()
    deform, diff, metric, adjrep, affine, lddmm = (m[k] for k in ("deform", "diff", "metric", "adjrep", "affine", "lddmm"))
    out = {}
    for dim, sp in ((2, (9, 8)), (3, (6, 7, 8))):
        for dtype, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            g = torch.Generator().manual_seed(100 + dim)
            key = f"d{dim}_{tag}_"
            I = torch.randn((2, 2) + sp, generator=g, dtype=torch.float64).to(dtype)
            u = (1.3 * torch.randn((2, dim) + sp, generator=g, dtype=torch.float64)).to(dtype)
            v = torch.randn((2, dim) + sp, generator=g, dtype=torch.float64).to(dtype)
            mm = torch.randn((2, dim) + sp, generator=g, dtype=torch.float64).to(dtype)
            go = torch.randn((2, 2) + sp, generator=g, dtype=torch.float64).to(dtype)
            out.update({key + "I": I, key + "u": u, key + "v": v, key + "m": mm, key + "go": go})
            # autograd plumbing of InterpFunction (deform.py:24-41), broadcast image included
            for bc in (False, True):
                Ii = (I[:1] if bc else I).clone().requires_grad_(True)
                ui = u.clone().requires_grad_(True)
                y = deform.interp(Ii, ui, dt=0.7)
                y.backward(go)
                out.update({key + f"interp_bc{int(bc)}": y.detach(), key + f"interp_bc{int(bc)}_dI": Ii.grad, key + f"interp_bc{int(bc)}_du": ui.grad})
            # diff.py wrappers (note the default displacement=True, diff.py:38)
            a, b = u.clone().requires_grad_(True), mm.clone().requires_grad_(True)
            y = diff.jacobian_times_vectorfield(a, b)
            y.backward(v)
            out.update({key + "jtv_default": y.detach(), key + "jtv_default_dv": a.grad, key + "jtv_default_dw": b.grad})
            a, b = u.clone().requires_grad_(True), mm.clone().requires_grad_(True)
            y = diff.jacobian_times_vectorfield_adjoint(a, b)
            y.backward(v)
            out.update({key + "jtv_adj": y.detach(), key + "jtv_adj_dv": a.grad, key + "jtv_adj_dw": b.grad})
            # FluidMetric: LUT construction (float32 rounding, metric.py:66-75) + FFT convention
            met = metric.FluidMetric([0.1, 0.05, 0.01])
            x = mm.clone().requires_grad_(True)
            s = met.sharp(x)
            s.backward(v)
            out.update({key + "sharp": s.detach(), key + "sharp_grad": x.grad, key + "flat": met.flat(mm).detach()})
            for d, (c, sn) in enumerate(zip(met.luts["cos"], met.luts["sin"])):
                out[key + f"lut_cos{d}"] = c
                out[key + f"lut_sin{d}"] = sn
            # adjoint representation (adjrep.py)
            small = 0.3 * u
            out[key + "ad"] = adjrep.ad(u, mm)
            out[key + "ad_star"] = adjrep.ad_star(u, mm)
            out[key + "Ad_star"] = adjrep.Ad_star(small, mm)
            out[key + "ad_dagger"] = adjrep.ad_dagger(u, mm, met)
            out[key + "Ad_dagger"] = adjrep.Ad_dagger(small, mm, met)
            out[key + "sym"] = adjrep.sym(u, mm, met)
            out[key + "sym_dagger"] = adjrep.sym_dagger(u, mm, met)
            # compositions (deform.py:53-70)
            out[key + "compose"] = deform.compose(u, v, ds=0.5, dt=-0.25)
            out[key + "compose_disp_vel"] = deform.compose_disp_vel(u, v, dt=-0.1)
            out[key + "compose_vel_disp"] = deform.compose_vel_disp(v, u, dt=0.2)
            # shooting (lddmm.py:20-105)
            met2 = metric.FluidMetric([0.1, 0.0, 0.01])
            m0 = (0.002 * mm).clone().requires_grad_(True)
            h = lddmm.expmap(met2, m0, num_steps=4)
            h.backward(v)
            out.update({key + "expmap4": h.detach(), key + "expmap4_grad": m0.grad})
            out[key + "expmap_advect3"] = lddmm.expmap_advect(met2, 0.002 * mm, num_steps=3).detach()
            out[key + "EPDiff_step"] = lddmm.EPDiff_step(met2, 0.002 * mm, 0.1, 0.2 * small).detach()
            mask = (torch.rand((1, 1) + sp, generator=g) > 0.3).to(dtype)
            out[key + "mask"] = mask
            out[key + "expmap2_masked"] = lddmm.expmap(met2, 0.002 * mm, num_steps=2, mommask=mask).detach()
            # affine.py: AffineInterpFunction plumbing, regrid (incl. displacement scaling), helpers
            A = (torch.eye(dim, dtype=torch.float64)[None] + 0.2 * torch.randn((2, dim, dim), generator=g, dtype=torch.float64)).to(dtype)
            T = (0.8 * torch.randn((2, dim), generator=g, dtype=torch.float64)).to(dtype)
            out.update({key + "A": A, key + "T": T})
            Ii, Ai, Ti = I.clone().requires_grad_(True), A.clone().requires_grad_(True), T.clone().requires_grad_(True)
            y = affine.affine_interp(Ii, Ai, Ti)
            y.backward(go)
            out.update({key + "affine": y.detach(), key + "affine_dI": Ii.grad, key + "affine_dA": Ai.grad, key + "affine_dT": Ti.grad})
            newshape = tuple(s_ + 3 for s_ in sp)
            ui = u.clone().requires_grad_(True)
            y = affine.regrid(ui, shape=newshape, displacement=True)
            gr = torch.randn(y.shape, generator=g, dtype=torch.float64).to(dtype)
            y.backward(gr)
            out.update({key + "regrid_disp": y.detach(), key + "regrid_disp_go": gr, key + "regrid_disp_grad": ui.grad})
            out[key + "regrid_plain"] = affine.regrid(I, shape=newshape).detach()
            Ainv, Tinv = affine.affine_inverse(A, T)
            out.update({key + "Ainv": Ainv, key + "Tinv": Tinv})
    arrays = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()}
    path = os.path.join(ROOT, "tests", "golden", "ref_python.npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}: {len(arrays)} arrays, {os.path.getsize(path)/1e3:.0f} kB")


if __name__ == "__main__":
    main()
