#!/usr/bin/env python3
"""Sweep LDS-splat tile configurations on the configs[1] workload (8 x 1x128^3 fp32)."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lagomorph_amd as lm
from bench import gaussian_blur, time_op

ext = lm.lagomorph_ext
dev = torch.device("cuda")
S, B = 128, 8
g = torch.Generator(device=dev).manual_seed(1234)
I = gaussian_blur(torch.randn((B, 1, S, S, S), device=dev, generator=g), 2.0); I = I / I.std()
u = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 8.0); u = u * (4.0 / u.abs().max())
go = torch.randn((B, 1, S, S, S), device=dev, generator=g)
V = B * S ** 3
ref = None
cfgs = []
for (tx, ty, tz) in ((16, 8, 64), (8, 8, 128), (16, 4, 128), (12, 8, 128), (8, 4, 128), (4, 8, 128), (8, 8, 64), (16, 8, 32),
                     (16, 16, 32), (32, 8, 32), (8, 8, 32), (4, 4, 128), (16, 16, 16), (32, 16, 16)):
    for (mx, mz) in ((1, 4), (1, 0), (0, 0), (2, 4)):
        for nt in (256, 512, 1024):
            cfgs.append((tx, ty, tz, mx, mx, mz, nt))
res = []
for cfg in cfgs:
    ext.set_splat_tile(*cfg)
    try:
        for need_u in (True, False):
            med, _ = time_op(lambda: ext.interp_backward(go, I, u, 1.0, True, need_u), reps=8, warm=2)
            res.append((med, cfg, need_u))
    except RuntimeError as e:
        print("cfg", cfg, "failed:", str(e)[:80])
res.sort()
for med, cfg, nu in res[:25]:
    print(f"{med*1e3:8.1f} us  need_u={nu!s:5}  tile={cfg}  {36.0*V/med/1e6 if nu else 24.0*V/med/1e6:7.0f} GB/s")
print("...")
for med, cfg, nu in res[-5:]:
    print(f"{med*1e3:8.1f} us  need_u={nu!s:5}  tile={cfg}")
