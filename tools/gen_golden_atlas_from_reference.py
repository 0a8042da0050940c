#!/usr/bin/env python3
"""Generate tests/golden/ref_atlas.npz by running the REFERENCE's own LDDMMAtlasBuilder
(/root/reference/lagomorph/lddmm.py:108-375, imported from where it lies -- nothing is copied) and
tests/golden/ref_affine_atlas.npz by running the reference's own `affine_atlas` (affine.py:288-415) and
`StandardizedDataset` (affine.py:418-438) on small seeded in-memory datasets, with the CPU oracle standing in for the CUDA-only `lagomorph_ext` exactly as
tools/gen_golden_from_reference.py does.

What these fixtures pin is the atlas builder's loop semantics above the extension boundary (SURVEY.md
section 8 rows f1 / f3): mean-image initialisation (data.py:308-336), `image_shape` regrid of I0,
`lddmm_steps` inner iterations with the image gradient taken on the last one only, `image_update_freq`
accumulation and the forced update at epoch end, `momentum_shape` != image shape (multiscale momenta:
regrid of the deformation and the rescaled regularisation term), momentum preconditioning, ragged last
minibatches, and the four loss histories.

The affine fixtures pin SURVEY section 8 row f4: mean-image initialisation through data.batch_average (or a given
I), `affine_steps` inner steps with the image gradient taken on the last one only, `image_update_freq` accumulation
incl. the reference's carry-over of a partial accumulation into the next epoch (affine.py:404-409 steps without
zeroing; :352-353 zeroes only when image_update_freq == 0 or in the first epoch), the A / T regularisers, ragged last
minibatches, the per-iteration and per-epoch losses, and the inverse-map resampling of StandardizedDataset.

Environment shims (none of them touches the reference's arithmetic): torch.rfft/irfft aliases, an empty
`h5py` module, and `Tensor.pin_memory` as the identity (no CUDA runtime in the build container).

    python tools/gen_golden_atlas_from_reference.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from gen_golden_from_reference import import_reference  # noqa: E402

CASES = {
    # name: (spatial shape, subjects, builder kwargs, I0 shape or None)
    "a3d": ((6, 6, 6), 4, dict(batch_size=2, num_epochs=3, lddmm_integration_steps=2, reg_weight=1e-1,
                               learning_rate_pose=2e-6, learning_rate_image=5e-2), None),
    "b2d_multiscale": ((8, 8), 5, dict(batch_size=2, num_epochs=3, lddmm_steps=2, lddmm_integration_steps=3,
                                       image_update_freq=2, reg_weight=5e-2, learning_rate_pose=5e-4,
                                       learning_rate_image=1e-1, momentum_shape=(5, 5),
                                       momentum_preconditioning=True), None),
    "c3d_multiscale_I0": ((6, 7, 8), 3, dict(batch_size=2, num_epochs=2, lddmm_integration_steps=2, reg_weight=1e-1,
                                            learning_rate_pose=2e-4, learning_rate_image=5e-2,
                                            momentum_shape=(4, 4, 5), image_update_freq=1), (5, 5, 6)),
}


def dataset(n, sp, seed):
    g = torch.Generator().manual_seed(seed)
    base = torch.randn((1,) + sp, generator=g, dtype=torch.float64)
    return [base + 0.3 * torch.randn((1,) + sp, generator=g, dtype=torch.float64) for _ in range(n)]


AFFINE_CASES = {
    # name: (spatial shape, subjects, affine_atlas kwargs, give I?)
    "a2d": ((12, 11), 5, dict(num_epochs=3, batch_size=2, learning_rate_A=2e-3, learning_rate_T=5e-2,
                              learning_rate_I=1.0), False),
    "b3d_freq_steps": ((7, 8, 6), 5, dict(num_epochs=3, batch_size=2, image_update_freq=2, affine_steps=2,
                                          reg_weightA=0.3, reg_weightT=0.05, learning_rate_A=3e-3,
                                          learning_rate_T=4e-2, learning_rate_I=0.7), False),
    "c2d_givenI_freq1": ((10, 9), 4, dict(num_epochs=2, batch_size=3, image_update_freq=1, affine_steps=3,
                                          reg_weightT=0.1, learning_rate_A=1e-3, learning_rate_T=2e-2,
                                          learning_rate_I=0.5), True),
    "d3d_f32": ((6, 7, 8), 4, dict(num_epochs=2, batch_size=2, learning_rate_A=2e-3, learning_rate_T=5e-2,
                                   learning_rate_I=1.0), False),
}


def affine_subjects(affine, n, sp, seed, dtype):
    """A smooth blob + texture pushed through small random affine maps (by the reference's own affine_interp)."""
    g = torch.Generator().manual_seed(seed)
    grids = torch.meshgrid(*[torch.arange(s, dtype=torch.float64) for s in sp], indexing="ij")
    c = [(s - 1) / 2 for s in sp]
    blob = torch.exp(-sum((gr - ci) ** 2 for gr, ci in zip(grids, c)) / (2 * (min(sp) / 4) ** 2))
    base = (blob + 0.1 * torch.randn(sp, generator=g, dtype=torch.float64))[None, None]
    d = len(sp)
    A = torch.eye(d, dtype=torch.float64)[None] + 0.08 * torch.randn((n, d, d), generator=g, dtype=torch.float64)
    T = 0.8 * torch.randn((n, d), generator=g, dtype=torch.float64)
    imgs = affine.affine_interp(base, A.contiguous(), T.contiguous()).detach()
    return [imgs[i].to(dtype) for i in range(n)]  # each (1, *sp), like one element of the reference's datasets


def main_affine(m):
    import importlib

    affine = m["affine"]
    data_mod = importlib.import_module("lagomorph.data")
    out = {}
    for seed, (name, (sp, n, kw, give_I)) in enumerate(AFFINE_CASES.items()):
        dtype = torch.float32 if name.endswith("f32") else torch.float64
        d = len(sp)
        subj = affine_subjects(affine, n, sp, 60 + seed, dtype)
        ds = data_mod.IndexedDataset(subj)
        As = torch.zeros((n, d, d), dtype=dtype)
        Ts = torch.zeros((n, d), dtype=dtype)
        I0 = None
        if give_I:
            I0 = torch.stack(subj).mean(dim=0) + 0.05 * torch.randn((1,) + sp, dtype=dtype,
                                                                     generator=torch.Generator().manual_seed(70 + seed))
            out[name + "_I0"] = I0.numpy()
        I, A2, T2, ep, it = affine.affine_atlas(ds, As, Ts, I=I0, loader_workers=0, gpu=None, **kw)
        out[name + "_data"] = torch.stack(subj).numpy()
        out[name + "_I"] = I.detach().numpy()
        out[name + "_A"] = A2.detach().numpy()
        out[name + "_T"] = T2.detach().numpy()
        out[name + "_epoch_losses"] = np.asarray(ep, dtype=np.float64)
        out[name + "_iter_losses"] = np.asarray(it, dtype=np.float64)
        # StandardizedDataset (affine.py:418-438) over the same subjects with the fitted parameters
        sd = affine.StandardizedDataset(subj, A2.detach(), T2.detach(), device="cpu")
        out[name + "_standardized"] = torch.stack([sd[i] for i in range(len(sd))]).detach().numpy()
        print(name, "epoch losses", out[name + "_epoch_losses"], "iters", len(it))
    path = os.path.join(ROOT, "tests", "golden", "ref_affine_atlas.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path)/1e3:.0f} kB")


def main():
    torch.Tensor.pin_memory = lambda self, *a, **k: self
    m = import_reference()
    if "--affine-only" in sys.argv:
        return main_affine(m)
    lddmm = m["lddmm"]
    out = {}
    for seed, (name, (sp, n, kw, i0sp)) in enumerate(CASES.items()):
        data = dataset(n, sp, 40 + seed)
        I0 = None
        if i0sp is not None:
            I0 = torch.randn((1, 1) + i0sp, generator=torch.Generator().manual_seed(90 + seed), dtype=torch.float64)
            out[name + "_I0"] = I0.numpy()
        b = lddmm.LDDMMAtlasBuilder(data, I0=I0, loader_workers=0, device="cpu", **kw)
        b.run()
        out[name + "_data"] = torch.stack(data).numpy()
        out[name + "_I"] = b.I.detach().numpy()
        out[name + "_ms"] = torch.cat([x.detach() for x in b.ms]).numpy()
        for k in ("epoch_losses", "epoch_reg_terms", "iter_losses", "iter_reg_terms"):
            out[name + "_" + k] = np.asarray(getattr(b, k), dtype=np.float64)
        print(name, "epoch losses", out[name + "_epoch_losses"])
    path = os.path.join(ROOT, "tests", "golden", "ref_atlas.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path)/1e3:.0f} kB")
    main_affine(m)


if __name__ == "__main__":
    main()
