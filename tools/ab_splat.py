#!/usr/bin/env python3
"""A/B timing helper: the splat cases of bench.py's micro section, several rounds (run once per library build)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import gaussian_blur, time_op

ext = lm.lagomorph_ext
dev = torch.device("cuda")
S, B = int(os.environ.get("S", 128)), int(os.environ.get("B", 8))
g = torch.Generator(device=dev).manual_seed(1234)
I = gaussian_blur(torch.randn((B, 1, S, S, S), device=dev, generator=g), 2.0)
u = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 8.0)
u = u * (4.0 / u.abs().max())
go = torch.randn((B, 1, S, S, S), device=dev, generator=g)
I3 = torch.randn((B, 3, S, S, S), device=dev, generator=g)
g3 = torch.randn((B, 3, S, S, S), device=dev, generator=g)
out = []
for r in range(3):
    a, _ = time_op(lambda: ext.interp_backward(go, I, u, 1.0, True, True), reps=30, warm=3)
    b, _ = time_op(lambda: ext.interp_backward(go, I, u, 1.0, True, False), reps=30, warm=3)
    c, _ = time_op(lambda: ext.interp_backward(g3, I3, u, 1.0, True, True), reps=30, warm=3)
    f, _ = time_op(lambda: ext.interp_forward(I, u, 1.0), reps=30, warm=3)
    out.append(f"C=1 d_I+d_u {a*1e3:6.1f}  d_I {b*1e3:6.1f}  C=3 {c*1e3:6.1f}  fwd {f*1e3:5.1f}")
print(sys.argv[1] if len(sys.argv) > 1 else "", " | ".join(out))
