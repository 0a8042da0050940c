#!/usr/bin/env python3
"""A/B in one process of lago_set_splat_shear_mc modes / tile settings (round 4 used it for the image-window, pipelined and
row-mapped splat kernels -- commits a5e8ce0 .. d7f13fb, removed again: profiles/r04_splat_pipeline.md), alternating
rounds, 30 warm-up + 30 timed launches each; configs[1] (8 x 1 x 128^3) and the three-channel reverse-sweep form at
128^3 / 160^3, unit and non-unit step, plus d_u bits and d_I agreement between the two.
env: CASES "S:B:C:dt,..." """
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import gaussian_blur, time_op

ext = lm.lagomorph_ext
dev = torch.device("cuda")
cases = [tuple(float(x) for x in c.split(":")) for c in
         os.environ.get("CASES", "128:8:1:1,128:8:3:1,128:8:3:-0.2,160:8:1:1,160:8:3:-0.2").split(",")]
# a mode may carry a tile setting: "5:8x8" = mode 5 with set_splat_shear(1, 8, 8, 0, 1, 1, 4, 1024)
MODES = [m for m in os.environ.get("MODES", "2,1").split(",")]


def apply(m):
    mode, _, tile = m.partition(":")
    tx, ty = (int(x) for x in tile.split("x")) if tile else (8, 6)
    ext.set_splat_shear(1, tx, ty, 0, 1, 1, 4, 1024)
    ext.set_splat_shear_mc(int(mode))


for S, B, C, dt in cases:
    S, B, C = int(S), int(B), int(C)
    g = torch.Generator(device=dev).manual_seed(1234)
    I = gaussian_blur(torch.randn((B, C, S, S, S), device=dev, generator=g), 2.0)
    I = I / I.std()
    u = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 8.0)
    u = u * (float(os.environ.get("AMP", 4.0)) / u.abs().max())
    go = torch.randn((B, C, S, S, S), device=dev, generator=g)
    res, rows = {}, {m: [] for m in MODES}
    for r in range(3):
        for m in MODES:
            apply(m)
            if r == 0:
                before = ext.path_launches()
                res[m] = ext.interp_backward(go, I, u, dt, True, True)
                after = ext.path_launches()
                res[(m, "path")] = [k for k in after if after[k] != before[k]]
            med, _ = time_op(lambda: ext.interp_backward(go, I, u, dt, True, True), reps=30, warm=30)
            rows[m].append(med * 1e3)
    ext.set_splat_shear(1, 8, 6, 0, 1, 1, 4, 1024)
    ext.set_splat_shear_mc(2)
    a, b = res[MODES[0]], res[MODES[-1]]
    same_du = torch.equal(a[1], b[1])
    dI_err = float((a[0] - b[0]).abs().max() / b[0].abs().max())
    alg = 4 * (3 * C + 6) * B * S ** 3
    print(f"S={S} B={B} C={C} dt={dt}: d_u same bits {same_du}, d_I rel diff {dI_err:.2e}")
    for m in MODES:
        best = min(rows[m])
        print(f"   mode {m:6s} {res[(m, 'path')]}: " + "  ".join(f"{x:7.1f}" for x in rows[m]) +
              f" us   best -> {alg / best / 1e6:.2f} TB/s = {alg / best / 8e6:.3f} of peak")
    del I, u, go, res
