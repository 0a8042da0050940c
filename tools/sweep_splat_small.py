#!/usr/bin/env python3
"""Compare a few LDS-splat tile configurations at C=1 and C=3 (batch 8 x 128^3), need_u on/off."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lagomorph_amd as lm
from bench import gaussian_blur, time_op
ext = lm.lagomorph_ext
dev = torch.device("cuda")
S, B = (int(sys.argv[1]) if len(sys.argv) > 1 else 128), 8
g = torch.Generator(device=dev).manual_seed(1234)
u = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 8.0); u = u * (4.0 / u.abs().max())
data = {}
for C in (1, 3):
    I = gaussian_blur(torch.randn((B, C, S, S, S), device=dev, generator=g), 2.0); I = I / I.std()
    data[C] = (I, torch.randn((B, C, S, S, S), device=dev, generator=g))
cfgs = [tuple(int(x) for x in a.split(",")) for a in sys.argv[2:]] or [
    (16, 8, 64, 1, 1, 4, 1024), (4, 8, 128, 1, 1, 4, 512), (8, 8, 64, 1, 1, 0, 512), (8, 8, 128, 1, 1, 4, 1024),
    (8, 4, 128, 1, 1, 4, 512), (8, 8, 32, 1, 1, 0, 512), (4, 4, 128, 1, 1, 4, 256), (8, 4, 64, 1, 1, 0, 256)]
for cfg in cfgs:
    ext.set_splat_tile(*cfg)
    row = []
    for C in (1, 3):
        I, go = data[C]
        for need_u in (True, False):
            med, _ = time_op(lambda: ext.interp_backward(go, I, u, 1.0, True, need_u), reps=10, warm=3)
            row.append(f"C={C} u={int(need_u)}: {med*1e3:6.1f}")
    print(cfg, "  ".join(row), flush=True)
