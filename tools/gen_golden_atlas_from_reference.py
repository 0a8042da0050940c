#!/usr/bin/env python3
"""Generate tests/golden/ref_atlas.npz by running the REFERENCE's own LDDMMAtlasBuilder
(/root/reference/lagomorph/lddmm.py:108-375, imported from where it lies -- nothing is copied) on small
seeded in-memory datasets, with the CPU oracle standing in for the CUDA-only `lagomorph_ext` exactly as
tools/gen_golden_from_reference.py does.

What these fixtures pin is the atlas builder's loop semantics above the extension boundary (SURVEY.md
section 8 rows f1 / f3): mean-image initialisation (data.py:308-336), `image_shape` regrid of I0,
`lddmm_steps` inner iterations with the image gradient taken on the last one only, `image_update_freq`
accumulation and the forced update at epoch end, `momentum_shape` != image shape (multiscale momenta:
regrid of the deformation and the rescaled regularisation term), momentum preconditioning, ragged last
minibatches, and the four loss histories.

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


def main():
    torch.Tensor.pin_memory = lambda self, *a, **k: self
    m = import_reference()
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


if __name__ == "__main__":
    main()
