#!/usr/bin/env python3
"""Profiling aid: interp forward / backward timings at configs[1] (batch 8 x 1x128^3) and C=3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lagomorph_amd as lm
from bench import gaussian_blur, time_op
ext = lm.lagomorph_ext
dev = torch.device("cuda")
S, B = 128, 8
g = torch.Generator(device=dev).manual_seed(1234)
for C in (1, 3):
    I = gaussian_blur(torch.randn((B, C, S, S, S), device=dev, generator=g), 2.0); I = I / I.std()
    u = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 8.0); u = u * (4.0 / u.abs().max())
    go = torch.randn((B, C, S, S, S), device=dev, generator=g)
    f, _ = time_op(lambda: ext.interp_forward(I, u, 1.0), reps=20, warm=5)
    b1, _ = time_op(lambda: ext.interp_backward(go, I, u, 1.0, True, True), reps=20, warm=5)
    b0, _ = time_op(lambda: ext.interp_backward(go, I, u, 1.0, True, False), reps=20, warm=5)
    print(f"C={C}: fwd {f*1e3:.1f} us  bwd(I,u) {b1*1e3:.1f} us  bwd(I only) {b0*1e3:.1f} us", flush=True)
