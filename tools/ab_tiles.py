#!/usr/bin/env python3
"""Careful A/B of sheared-window tile settings: alternating rounds, 30 warm-up + 30 timed launches each.
env: S (128), B (8), C (3), TILES (4x8x0,8x6x0,...)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import gaussian_blur, time_op

ext = lm.lagomorph_ext
dev = torch.device("cuda")
S, B, C = int(os.environ.get("S", 128)), int(os.environ.get("B", 8)), int(os.environ.get("C", 3))
# TX x TY x TZ [x threads [x mz]]
tiles = [tuple(int(x) for x in t.split("x")) for t in os.environ.get("TILES", "4x8x0,8x6x0,5x10x0").split(",")]
MC = int(os.environ.get("MC", 2))
ext.set_splat_shear_mc(MC)
g = torch.Generator(device=dev).manual_seed(1234)
I = gaussian_blur(torch.randn((B, C, S, S, S), device=dev, generator=g), 2.0)
u = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 8.0)
u = u * (3.0 / u.abs().max())
go = torch.randn((B, C, S, S, S), device=dev, generator=g)
rows = {t: [] for t in tiles}
for r in range(3):
    for t in tiles:
        ext.set_splat_shear(1, tx=t[0], ty=t[1], tz=t[2], mx=1, my=1, mz=t[4] if len(t) > 4 else 4,
                            nthreads=t[3] if len(t) > 3 else 1024)
        med, _ = time_op(lambda: ext.interp_backward(go, I, u, 1.0, True, True), reps=30, warm=30)
        rows[t].append(med * 1e3)
print(f"S={S} B={B} C={C}")
for t in tiles:
    print(f"  {'x'.join(str(x) for x in t)}: " + "  ".join(f"{x:7.1f}" for x in rows[t]))
