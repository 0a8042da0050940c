#!/usr/bin/env python3
"""Compare LDS-staged gather tile configurations against the direct kernels (batch 8 x 3x128^3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lagomorph_amd as lm
from bench import gaussian_blur, time_op
ext = lm.lagomorph_ext
dev = torch.device("cuda")
S, B = 128, 8
g = torch.Generator(device=dev).manual_seed(99)
sh = (B, 3, S, S, S)
v, w = (torch.randn(sh, device=dev, generator=g) for _ in range(2))
u = gaussian_blur(torch.randn(sh, device=dev, generator=g), 8.0)
u = u * (4.0 / u.abs().max())
I1 = torch.randn((B, 1, S, S, S), device=dev, generator=g)


def row(tag):
    a, _ = time_op(lambda: ext.compose(u, v, -0.1, 1.0), reps=10, warm=3)
    b, _ = time_op(lambda: ext.Ad_star(u, w), reps=10, warm=3)
    c, _ = time_op(lambda: ext.interp_forward(I1, u, 1.0), reps=10, warm=3)
    print(f"{tag:40s} compose(ds=-0.1) {a*1e3:7.1f}  ad_star {b*1e3:7.1f}  interp C=1 {c*1e3:7.1f} us", flush=True)


ext.set_gather_mode(0)
row("direct")
ext.set_gather_mode(1)
cfgs = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [
    (8, 8, 64, 2, 2, 2, 1024), (8, 8, 64, 3, 3, 3, 1024), (8, 8, 64, 4, 4, 4, 1024), (8, 8, 64, 1, 1, 1, 1024),
    (8, 8, 64, 2, 2, 2, 512), (16, 8, 32, 2, 2, 2, 1024), (8, 16, 32, 3, 3, 3, 1024), (4, 8, 128, 3, 3, 0, 1024),
    (16, 16, 16, 3, 3, 3, 1024)]
for cfg in cfgs:
    ext.set_gather_tile(*cfg)
    row(str(cfg))
