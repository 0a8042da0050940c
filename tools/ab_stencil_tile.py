#!/usr/bin/env python3
"""A/B in one process of the LDS row-tile stencil kernels (csrc/stencil_tile.hpp) against the direct kernels:
Ad_star (with / without the saved resampled momentum) and jacobian_times_vectorfield_backward, bit compare + timing.
env: S (128), B (32)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import gaussian_blur, time_op

ext = lm.lagomorph_ext
S, B = int(os.environ.get("S", 128)), int(os.environ.get("B", 32))
shape = tuple(int(x) for x in os.environ["SHAPE"].split("x")) if "SHAPE" in os.environ else (S, S, S)
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(5)
phi = gaussian_blur(torch.randn((B, 3) + shape, device=dev, generator=g), 6.0)
phi = phi * (4.0 / phi.abs().max())
m = torch.randn((B, 3) + shape, device=dev, generator=g)
go = torch.randn((B, 3) + shape, device=dev, generator=g)
V = B * shape[0] * shape[1] * shape[2]
cases = {
    "Ad_star": (lambda: ext.Ad_star(phi, m), 36),
    "Ad_star(save)": (lambda: ext.Ad_star(phi, m, save_resampled=True)[0], 48),
    "jtv_backward": (lambda: ext.jacobian_times_vectorfield_backward(go, phi, m, True, False, True, True)[0], 60),
}
ref = {}
for r in range(2):
    for tile in [int(x) for x in os.environ.get('TILES', '0,1').split(',')]:
        ext.set_stencil_tile(tile)
        for name, (fn, bpv) in cases.items():
            out = fn()
            same = "first" if name not in ref else ("same" if torch.equal(out, ref[name]) else "DIFFER")
            ref.setdefault(name, out)
            t, _ = time_op(fn, reps=30, warm=20)
            print(f"{shape} B={B} tile={tile} {name:16s}: {t*1e3:8.1f} us  {bpv*V/t/1e9:6.2f} TB/s alg  bits {same}", flush=True)
ext.set_stencil_tile(1)
